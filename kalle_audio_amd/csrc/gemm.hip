// bf16 MFMA GEMM for the DiT block (gfx950): C[M,N] = op(A)[M,K] * op(B)[K,N] (+ fused epilogue)
//
// Replaces the torch ops behind the reference's nn.Linear calls on the DiT path
// (stable_audio_tools/models/transformer.py:216,252,411,414,419,541,774,807) and their autograd
// backward (dgrad / wgrad), which the reference leaves to torch.
//
// Operand storage (both operands, independently):
//   "k-contiguous": X[R][K] row-major (activations [M][K]; nn.Linear weights [N][K])
//   "k-major":      X[K][R] row-major (dgrad: weights [N'][K'] contracted over N';
//                                       wgrad: dY[M][N'] and X[M][K'] contracted over M)
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 as 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Global->register->LDS staging, double-buffered LDS, one barrier per K-tile.
//   k-contiguous image: [128][64] bf16, 128-B rows, 16-B chunk index XOR ((row>>1)&7)  -> ds_read_b128 conflict-free
//   k-major image:      [64][128] bf16, 288-B row stride, odd 8-row k-blocks stored with rows 0-3 <-> 4-7 swapped
//                       -> ds_read_b64_tr_b16 (hardware transpose read) conflict-free
// Epilogue: accumulators -> LDS fp32 tile -> coalesced 16-B row stores with bias / adaLN gate / residual / accumulate.
#include "gemm_common.h"
#include "../../include/kalle_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int KM_STRIDE = 288;            // bytes per k-row of a k-major LDS image (256 + 32 pad)
constexpr int KC_BYTES = 128 * 128;       // k-contiguous image
constexpr int KM_BYTES = 64 * KM_STRIDE;  // k-major image
constexpr int EP_LD = 132;                // fp32 epilogue tile leading dim (floats)


// ---- staging: global -> registers ------------------------------------------------------------
template <bool KM>
__device__ __forceinline__ void stage_load(i32x4 (&r)[4], const bf16_t* X, int64_t ld, int R, int K,
                                           int r0, int k0, int tid) {
    if constexpr (!KM) {
        // tile = 128 rows x 64 k; 8 chunks (16 B) per row
        const int vrows = min(128, R - r0);
        const bf16_t* base = X + (int64_t)r0 * ld;
        const uint32_t recs = (uint32_t)(((int64_t)(vrows - 1) * ld + K) * 2);
        __amdgpu_buffer_rsrc_t rs = make_rsrc(base, recs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i;
            const int row = id >> 3, c = id & 7;
            const int k = k0 + 8 * c;
            uint32_t off = (uint32_t)(((int64_t)row * ld + k) * 2);
            off = (k < K) ? off : 0xFFFFFFF0u;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
    } else {
        // tile = 64 k-rows x 128 r; 16 chunks per k-row
        const int vk = min(64, K - k0);
        const bf16_t* base = X + (int64_t)k0 * ld + r0;
        const uint32_t recs = (uint32_t)(((int64_t)(vk - 1) * ld + (R - r0)) * 2);
        __amdgpu_buffer_rsrc_t rs = make_rsrc(base, recs);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + 256 * i;
            const int krow = id >> 4, c = id & 15;
            uint32_t off = (uint32_t)(((int64_t)krow * ld + 8 * c) * 2);
            off = (r0 + 8 * c < R) ? off : 0xFFFFFFF0u;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
    }
}

// ---- staging: registers -> LDS ---------------------------------------------------------------
template <bool KM>
__device__ __forceinline__ void stage_write(const i32x4 (&r)[4], char* s, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = tid + 256 * i;
        int addr;
        if constexpr (!KM) {
            const int row = id >> 3, c = id & 7;
            addr = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        } else {
            const int krow = id >> 4, c = id & 15;
            const int pos = (krow & ~7) | ((krow & 7) ^ (((krow >> 3) & 1) << 2));
            addr = pos * KM_STRIDE + c * 16;
        }
        *reinterpret_cast<i32x4*>(s + addr) = r[i];
    }
}

// ---- fragment read: 16 (rows of the operand) x 32 (k) for MFMA 16x16x32 -----------------------
// rbase: first of the 16 operand rows inside the 128-row tile; s: k-step (0/1) inside the 64-deep tile
template <bool KM>
__device__ __forceinline__ bf16x8 frag_read(const char* sm, int rbase, int s, int lane) {
    if constexpr (!KM) {
        const int row = rbase + (lane & 15);
        const int c = 4 * s + (lane >> 4);
        const int addr = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
        return *reinterpret_cast<const bf16x8*>(sm + addr);
    } else {
        const int g = lane >> 4, i = lane & 15;
        const int q = i >> 2, pp = i & 3;
        const int b = 4 * s + g;  // 8-row k-block
        const int odd = b & 1;
        const int row1 = 8 * b + (odd ? 4 : 0) + q;
        const int row2 = 8 * b + (odd ? 0 : 4) + q;
        const int cb = (rbase + 4 * pp) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sm + row1 * KM_STRIDE + cb));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, sm + row2 * KM_STRIDE + cb));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <bool A_KM, bool B_KM, bool C_F32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = A_KM ? KM_BYTES : KC_BYTES;
    constexpr int B_BYTES = B_KM ? KM_BYTES : KC_BYTES;
    constexpr int BUF = A_BYTES + B_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int tm, tn;
    gemm_tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    i32x4 ra[4], rb[4];
    const int nk = (p.K + BK - 1) / BK;

    stage_load<A_KM>(ra, p.A, p.lda, p.M, p.K, m0, 0, tid);
    stage_load<B_KM>(rb, p.B, p.ldb, p.N, p.K, n0, 0, tid);
    stage_write<A_KM>(ra, smem, tid);
    stage_write<B_KM>(rb, smem + A_BYTES, tid);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            stage_load<A_KM>(ra, p.A, p.lda, p.M, p.K, m0, (kt + 1) * BK, tid);
            stage_load<B_KM>(rb, p.B, p.ldb, p.N, p.K, n0, (kt + 1) * BK, tid);
        }
        const char* sa = smem + cur * BUF;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = frag_read<A_KM>(sa, wm * 64 + 16 * t, s, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) bfr[t] = frag_read<B_KM>(sb, wn * 64 + 16 * t, s, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
        }
        if (more) {
            char* na = smem + (cur ^ 1) * BUF;
            stage_write<A_KM>(ra, na, tid);
            stage_write<B_KM>(rb, na + A_BYTES, tid);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: accumulators -> LDS fp32 tile (C layout: col = lane&15, row = 4*(lane>>4)+reg) ----
    float* et = reinterpret_cast<float*>(smem);
    {
        const int g = lane >> 4, c = lane & 15;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    et[(wm * 64 + 16 * mt + 4 * g + r) * EP_LD + wn * 64 + 16 * nt + c] = acc[mt][nt][r];
    }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int id = tid + 256 * i;
        const int row = id >> 4, cc = id & 15;
        const int gm = m0 + row, gn = n0 + 8 * cc;
        if (gm >= p.M || gn >= p.N) continue;
        const int64_t crow = p.c_rpb > 0 ? (int64_t)(gm / p.c_rpb) * p.c_brows + p.c_roff + gm % p.c_rpb : gm;
        float v[8];
        {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(et + row * EP_LD + 8 * cc);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(et + row * EP_LD + 8 * cc + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = v0[j] * p.alpha; v[4 + j] = v1[j] * p.alpha; }
        }
        if (p.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + gn);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
        }
        if (p.gate) {
            // adaLN gating, transformer.py:667,681: x * sigmoid(1 - gate)
            const float* gp = p.gate + (int64_t)(gm / p.rows_per_batch) * p.ldg + gn;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] *= sigmoidf_(1.f - g0[j]); v[4 + j] *= sigmoidf_(1.f - g1[j]); }
        }
        if (p.row_mask && !p.row_mask[gm]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        if (p.residual) {
            const float* rp = p.residual + crow * p.ldr + gn;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
        }
        if constexpr (C_F32) {
            float* cp = reinterpret_cast<float*>(p.C) + crow * p.ldc + gn;
            if (p.accumulate) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp);
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(cp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += c0[j]; v[4 + j] += c1[j]; }
            }
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
            bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gn;
            i32x4 o;
            o[0] = (int)pack_bf16x2(v[0], v[1]);
            o[1] = (int)pack_bf16x2(v[2], v[3]);
            o[2] = (int)pack_bf16x2(v[4], v[5]);
            o[3] = (int)pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<i32x4*>(cp) = o;
        }
    }
}

template <bool A_KM, bool B_KM, bool C_F32>
int launch(const GemmParams& p, hipStream_t st) {
    constexpr int A_BYTES = A_KM ? KM_BYTES : KC_BYTES;
    constexpr int B_BYTES = B_KM ? KM_BYTES : KC_BYTES;
    constexpr int pipe = 2 * (A_BYTES + B_BYTES);
    constexpr int epi = BM * EP_LD * 4;
    constexpr int lds = pipe > epi ? pipe : epi;
    static std::atomic<uint64_t> lds_ok{0};
    kalle_allow_lds(reinterpret_cast<const void*>(gemm_bf16_kernel<A_KM, B_KM, C_F32>), lds, lds_ok);
    dim3 grid(p.tiles_m * p.tiles_n), block(256);
    KALLE_LAUNCH((gemm_bf16_kernel<A_KM, B_KM, C_F32>), grid, block, lds, st, p);
    return kalle_check_launch();
}

}  // namespace

int kalle_gemm_v1_launch(const GemmParams& pin, bool a_km, bool b_km, bool f32, hipStream_t st) {
    GemmParams p = pin;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.group_m = p.tiles_m < 8 ? p.tiles_m : 8;
    if (!a_km && !b_km) return f32 ? launch<false, false, true>(p, st) : launch<false, false, false>(p, st);
    if (!a_km && b_km) return f32 ? launch<false, true, true>(p, st) : launch<false, true, false>(p, st);
    if (a_km && !b_km) return f32 ? launch<true, false, true>(p, st) : launch<true, false, false>(p, st);
    return f32 ? launch<true, true, true>(p, st) : launch<true, true, false>(p, st);
}
