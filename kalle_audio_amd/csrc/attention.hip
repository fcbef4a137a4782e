// Fused attention for the DiT block and the Llama decoder layers (gfx950), forward and backward, head dim 64,
// optional causal mask (keys j <= i + Nk - Nq), optional key-padding mask, rotary on the first 32 or on all 64 dims.
// Reference: stable_audio_tools/models/transformer.py:396-547 (Attention.forward: rotary 430-444, key mask 446-462,
// softmax(QK^T/sqrt(d))V 502-530 / SDPA 382-387, GQA repeat_interleave 337-340, 505-508) and
// transformer.py:146-170 (rotate_half / apply_rotary_pos_emb, partial rotary on the first 32 dims).
//
// Design (sequences of ~126 latent frames at 12.5 Hz, S~130 context tokens): one workgroup of 4 waves owns a
// 128-row block of one (batch, head); the whole 128-key K/V block is LDS-resident (one image per tensor,
// [128][64] bf16 with a 160-B row stride that is conflict-free for BOTH ds_read_b128 row fragments and
// ds_read_b64_tr_b16 transposed fragments), longer sequences loop over blocks with an online softmax.
// Q/K/V are read in place from the projection outputs ([B][N][ld] with a column offset per head): no head
// transposes ever touch HBM. RoPE is applied while staging Q/K into LDS (fp32 math, as the reference does) and
// un-applied on dQ/dK in registers.
// MFMA orientation: scores are computed TRANSPOSED (S^T = K Q^T) so that the softmax row statistics of a query
// live in one lane (+2 shuffles) and the fp32 score accumulators are, after bf16 packing, directly the B operand
// of the next MFMA (O^T = V^T P^T) - nothing but V^T (a hardware transposed LDS read) crosses lanes.
#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

constexpr int AT_STRIDE = 160;             // bytes per LDS row: 64 bf16 + 32 B pad
constexpr int AT_TILE = 128 * AT_STRIDE;   // 20480 B
constexpr float SM_SCALE = 0.125f;         // 1/sqrt(64)
constexpr float NEG_BIG = -1.0e30f;
constexpr float SM_SCALE_LOG2E = SM_SCALE * 1.4426950408889634f;   // exponent of 2 per unit of raw score
constexpr float LOG2E = 1.4426950408889634f;
constexpr float M_INIT = -2.0e30f;         // running max before the first block: below a fully masked row's NEG_BIG

// stage a [128][64] bf16 tile (rows row0.., `nvalid` valid) into LDS, optionally applying rotary on the first `rot`
// (32: DiT partial rotary; 64: Llama) dims: out = x cos + rotate_half(x) sin, tables [pos][rot/2].
// Split in two so that a kernel can put the global loads of several tiles in flight together (one HBM latency instead
// of one per tile) and do the rotary + LDS writes afterwards.
// A thread owns ONE row and TWO 8-element chunks of it: the two partners of a rotary pair (chunk j and j + rot/16), so the
// rotation needs no second load of the partner and one fetch of the cos / sin row per pair (a quarter of the table traffic
// and half the data traffic of one-chunk-per-thread staging); chunks outside the rotary dims are plain adjacent pairs.
template <int NT>
struct TileRegs {
    static_assert(NT == 512, "128 rows x 4 chunk pairs");
    i32x4 v[2];
};
__device__ __forceinline__ void tile_chunks(int rot, int j, int& ca, int& cb) {
    const int hc = rot >> 4;                            // 8-element chunks per rotary half: 0, 2 (DiT) or 4 (Llama)
    if (j < hc) { ca = j; cb = j + hc; }                // a rotary pair
    else if (hc == 2) { ca = 2 * j; cb = 2 * j + 1; }   // rot 32: j = 2, 3 -> chunks (4, 5), (6, 7) pass through
    else { ca = 2 * j; cb = 2 * j + 1; }                // rot 0
}
template <int NT>
__device__ __forceinline__ void tile_load(TileRegs<NT>& t, const bf16_t* src, int64_t ld, int row0, int nvalid, int rot,
                                          int tid) {
    const int row = tid >> 2;
    int ca, cb;
    tile_chunks(rot, tid & 3, ca, cb);
    t.v[0] = i32x4{0, 0, 0, 0};
    t.v[1] = i32x4{0, 0, 0, 0};
    if (row < nvalid) {
        const bf16_t* rp = src + (int64_t)(row0 + row) * ld;
        t.v[0] = *reinterpret_cast<const i32x4*>(rp + 8 * ca);
        t.v[1] = *reinterpret_cast<const i32x4*>(rp + 8 * cb);
    }
}
// The rotary factors of a thread's pair, fetched WITH the tile (rope_load next to tile_load) instead of inside tile_store: there
// the table loads start only once the tile data has arrived - a second, dependent memory latency in front of the first barrier
// (the round-3 stamps: 4.9 us of load + stage per self-attention workgroup against 2.8 us for the rotary-free cross-attention).
struct RopeRegs { f32x4 c0, c1, s0, s1; };
__device__ __forceinline__ RopeRegs rope_load(const float* __restrict__ cosT, const float* __restrict__ sinT, int row0,
                                              int nvalid, int rot, int tid, int pos_off = 0) {
    const int row = tid >> 2, j = tid & 3;
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = c0, s0 = c0, s1 = c0;
    if (row < nvalid && j < (rot >> 4)) {
        const float* cp = cosT + (int64_t)(row0 + row + pos_off) * (rot >> 1) + j * 8;
        const float* sp = sinT + (int64_t)(row0 + row + pos_off) * (rot >> 1) + j * 8;
        c0 = *reinterpret_cast<const f32x4*>(cp); c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        s0 = *reinterpret_cast<const f32x4*>(sp); s1 = *reinterpret_cast<const f32x4*>(sp + 4);
    }
    return RopeRegs{c0, c1, s0, s1};
}

template <int NT>
__device__ __forceinline__ void tile_store(char* lds, const TileRegs<NT>& t, int row0, int nvalid,
                                           const float* __restrict__ cosT, const float* __restrict__ sinT, int rot,
                                           int tid, int pos_off, bool use_pre, const RopeRegs pre) {
    const int row = tid >> 2, j = tid & 3;
    int ca, cb;
    tile_chunks(rot, j, ca, cb);
    i32x4 va = t.v[0], vb = t.v[1];
    if (row < nvalid && j < (rot >> 4)) {               // out_a = a cos - b sin, out_b = b cos + a sin (rotate_half)
        f32x4 c0, c1, s0, s1;
        if (use_pre) {
            c0 = pre.c0; c1 = pre.c1; s0 = pre.s0; s1 = pre.s1;
        } else {
            const float* cp = cosT + (int64_t)(row0 + row + pos_off) * (rot >> 1) + j * 8;
            const float* sp = sinT + (int64_t)(row0 + row + pos_off) * (rot >> 1) + j * 8;
            c0 = *reinterpret_cast<const f32x4*>(cp); c1 = *reinterpret_cast<const f32x4*>(cp + 4);
            s0 = *reinterpret_cast<const f32x4*>(sp); s1 = *reinterpret_cast<const f32x4*>(sp + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf16lo((uint32_t)va[e]), a1 = bf16hi((uint32_t)va[e]);
            const float b0 = bf16lo((uint32_t)vb[e]), b1 = bf16hi((uint32_t)vb[e]);
            const float cc0 = e < 2 ? c0[2 * e] : c1[2 * e - 4], cc1 = e < 2 ? c0[2 * e + 1] : c1[2 * e - 3];
            const float ss0 = e < 2 ? s0[2 * e] : s1[2 * e - 4], ss1 = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
            va[e] = (int)pack_bf16x2(a0 * cc0 - b0 * ss0, a1 * cc1 - b1 * ss1);
            vb[e] = (int)pack_bf16x2(b0 * cc0 + a0 * ss0, b1 * cc1 + a1 * ss1);
        }
    }
    *reinterpret_cast<i32x4*>(lds + row * AT_STRIDE + 16 * ca) = va;
    *reinterpret_cast<i32x4*>(lds + row * AT_STRIDE + 16 * cb) = vb;
}
template <int NT>
__device__ __forceinline__ void tile_store(char* lds, const TileRegs<NT>& t, int row0, int nvalid,
                                           const float* __restrict__ cosT, const float* __restrict__ sinT, int rot,
                                           int tid, int pos_off = 0) {
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    tile_store<NT>(lds, t, row0, nvalid, cosT, sinT, rot, tid, pos_off, false, RopeRegs{z, z, z, z});
}
template <int NT>
__device__ __forceinline__ void stage_tile(char* lds, const bf16_t* src, int64_t ld, int row0, int nvalid,
                                           const float* __restrict__ cosT, const float* __restrict__ sinT, int rot,
                                           int tid, int pos_off = 0) {
    TileRegs<NT> t;
    tile_load<NT>(t, src, ld, row0, nvalid, rot, tid);
    tile_store<NT>(lds, t, row0, nvalid, cosT, sinT, rot, tid, pos_off);
}

// fragment of 16 tile rows (rbase..) x 32 d (k-step s): MFMA operand whose k index is the head dim
__device__ __forceinline__ bf16x8 rowfrag(const char* lds, int rbase, int s, int lane) {
    const int row = rbase + (lane & 15);
    const int c = 4 * s + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + row * AT_STRIDE + 16 * c);
}
// transposed fragment: MFMA operand whose k index is the TILE ROW and whose lane index is d (dbase + lane&15).
// element j<4 -> row ra + 4g + j ; j>=4 -> row rb + 4g + (j-4): the k order of a packed accumulator pair.
__device__ __forceinline__ bf16x8 trfrag(const char* lds, int ra, int rb, int dbase, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int qq = i >> 2, pp = i & 3;
    const int cb = (dbase + 4 * pp) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + (ra + 4 * g + qq) * AT_STRIDE + cb));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + (rb + 4 * g + qq) * AT_STRIDE + cb));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
    i32x4 w;
    w[0] = (int)pack_bf16x2(a[0], a[1]);
    w[1] = (int)pack_bf16x2(a[2], a[3]);
    w[2] = (int)pack_bf16x2(b[0], b[1]);
    w[3] = (int)pack_bf16x2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ float xor16_32_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor16_32_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// raw VALU forms: hipcc guards fmaxf / fminf of MFMA results with a canonicalising v_max x, x per operand (signalling-NaN
// semantics) and only packs some of the fp32 pairs; in the softmax sections those extra instructions cost as much as the
// exponentials themselves.
// NB the compiler's hazard recognizer does not look inside inline asm: an MFMA result must not be read by one of these
// within the MFMA's pass count + 3 cycles.  mfma_results_ready() is the fence between the last MFMA and the first raw use.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mfma_results_ready() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 lo2(const f32x4& v) { return f32x2{v[0], v[1]}; }
__device__ __forceinline__ f32x2 hi2(const f32x4& v) { return f32x2{v[2], v[3]}; }

struct AttnParams {
    const bf16_t* q; int64_t ldq; int q_off;
    const bf16_t* k; int64_t ldk; int k_off;
    const bf16_t* v; int64_t ldv; int v_off;
    bf16_t* out; int64_t ldo;
    float* lse;
    const float* cosT; const float* sinT; int rot;
    const uint8_t* mask;
    int B, H, Hkv, Nq, Nk;
    int causal;          // keys j <= i + (Nk - Nq) only
    int qpos;            // rotary position of query row 0 (Nk - Nq when causal: queries are the LAST Nq positions, KV cache)
    // backward only
    const bf16_t* dout; float* delta;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    unsigned long long* stamps;   // diagnostics (kalle_attn_debug_stamps): [workgroup][8] s_memrealtime stamps of wave 0, or NULL
};
__device__ __forceinline__ void attn_stamp(const AttnParams& p, int tid, int idx) {
    if (p.stamps && tid == 0)
        p.stamps[((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * 8 + idx] = __builtin_amdgcn_s_memrealtime();
}

// ================================================================================================ forward
// QT = 16-row query tiles per wave: 2 -> 4 waves per workgroup (wave = 32 queries), 1 -> 8 waves (wave = 16 queries,
// <= 128 VGPRs so two workgroups = 16 waves share a CU and hide each other's load -> LDS -> MFMA latency chain)
// ---- stores of an owner row's 64 features from the MFMA result layout ------------------------------------------------------------
// Lane (g, li) holds features 4g .. 4g + 3 of each 16-feature block dt of row li: straight from that layout a lane writes 8 bytes
// per block - 32 contiguous bytes per row and store instruction.  Lanes g and g ^ 1 trade blocks instead (v_permlane16_swap_b32:
// the odd 16-lane rows of one register against the even rows of the other, one instruction per traded dword pair), so that each
// owns 8 consecutive features of two blocks: 16-byte stores, 64 contiguous bytes per row and instruction, half the store
// instructions (the GEMM epilogues' lesson: the store side is bound by the number of store instructions, not by bytes).
// `row`: the row's first feature (no lane offset).  Partners share li, i.e. the row and its validity: safe under a row guard.
#ifndef KALLE_ATTN_ST16
#define KALLE_ATTN_ST16 1
#endif
__device__ __forceinline__ void store_row64(bf16_t* row, const f32x4& v0, const f32x4& v1, const f32x4& v2, const f32x4& v3, int g) {
    const f32x4 v[4] = {v0, v1, v2, v3};
    uint32_t w[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { w[dt][0] = pack_bf16x2(v[dt][0], v[dt][1]); w[dt][1] = pack_bf16x2(v[dt][2], v[dt][3]); }
#if KALLE_ATTN_ST16
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        // even g keeps block 2 pr and receives the partner's; odd g keeps block 2 pr + 1 and receives the partner's (lower features)
        const auto a = __builtin_amdgcn_permlane16_swap(w[2 * pr][0], w[2 * pr + 1][0], false, false);
        const auto c = __builtin_amdgcn_permlane16_swap(w[2 * pr][1], w[2 * pr + 1][1], false, false);
        *reinterpret_cast<i32x4*>(row + 16 * (2 * pr + (g & 1)) + 4 * (g & ~1)) = i32x4{(int)a[0], (int)c[0], (int)a[1], (int)c[1]};
    }
#else
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<i32x2*>(row + 16 * dt + 4 * g) = i32x2{(int)w[dt][0], (int)w[dt][1]};
#endif
}


template <int QT>
__global__ __launch_bounds__(128 / (16 * QT) * 64, QT == 1 ? 4 : 2) void attn_fwd_kernel(AttnParams p) {
    constexpr int NT = 128 / (16 * QT) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Ks = smem + AT_TILE;
    char* Vs = smem + 2 * AT_TILE;
    float* kbias = reinterpret_cast<float*>(smem + 3 * AT_TILE);  // [128] + [32] for the folded tail
    char* Kt = smem + 3 * AT_TILE + 640;                          // tail keys (fold_tail): two 16-row tiles of K and of V
    char* Vt = Kt + 32 * AT_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    // causal: the last query block streams the most key blocks - dispatch the heavy workgroups first
    const int b = blockIdx.z, h = blockIdx.y, q0 = (p.causal ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * 128;
    // The DiT's cross-attention has 130 context tokens: the 2 keys beyond the first block used to cost a second pass through
    // the staging code - barrier, global loads with nothing to hide their latency, barrier - 16 us of a 111-us call.  A tail of
    // up to 32 keys (no rotary, no causal mask) is fetched WITH the first block into its own small tiles instead and multiplied
    // right behind it: no further barrier, no exposed load.
    const bool fold_tail = QT == 1 && !p.causal && p.rot == 0 && p.Nk > 128 && p.Nk <= 160;
    attn_stamp(p, tid, 0);
    const int hk = h / (p.H / p.Hkv);

    const bf16_t* qsrc = p.q + (int64_t)b * p.Nq * p.ldq + p.q_off + h * 64;
    const bf16_t* ksrc = p.k + (int64_t)b * p.Nk * p.ldk + p.k_off + hk * 64;
    const bf16_t* vsrc = p.v + (int64_t)b * p.Nk * p.ldv + p.v_off + hk * 64;

    {   // Q and the first K / V block: all global loads in flight together, then rotary + LDS writes
        TileRegs<NT> tq, tk, tv;
        const int kv0 = min(128, p.Nk);
        tile_load<NT>(tq, qsrc, p.ldq, q0, min(128, p.Nq - q0), p.rot, tid);
        tile_load<NT>(tk, ksrc, p.ldk, 0, kv0, p.rot, tid);
        tile_load<NT>(tv, vsrc, p.ldv, 0, kv0, 0, tid);
        const RopeRegs rq = rope_load(p.cosT, p.sinT, q0, min(128, p.Nq - q0), p.rot, tid, p.qpos);   // (rot == 0: no loads)
        const RopeRegs rk = rope_load(p.cosT, p.sinT, 0, kv0, p.rot, tid);
        i32x4 tt = i32x4{0, 0, 0, 0};                   // one 16-byte chunk of the tail per thread: K rows then V rows (zero beyond Nk)
        if constexpr (QT == 1) {
            if (fold_tail) {
                const int row = (tid >> 3) & 31, key = 128 + row;
                if (key < p.Nk)
                    tt = *reinterpret_cast<const i32x4*>((tid < 256 ? ksrc + (int64_t)key * p.ldk : vsrc + (int64_t)key * p.ldv) +
                                                         8 * (tid & 7));
            }
        }
        tile_store<NT>(Qs, tq, q0, min(128, p.Nq - q0), p.cosT, p.sinT, p.rot, tid, p.qpos, p.rot != 0, rq);
        tile_store<NT>(Ks, tk, 0, kv0, p.cosT, p.sinT, p.rot, tid, 0, p.rot != 0, rk);
        tile_store<NT>(Vs, tv, 0, kv0, nullptr, nullptr, 0, tid);
        if constexpr (QT == 1) {
            if (fold_tail) {
                *reinterpret_cast<i32x4*>((tid < 256 ? Kt : Vt) + ((tid >> 3) & 31) * AT_STRIDE + 16 * (tid & 7)) = tt;
                if (tid < 32) {
                    float bias = 0.f;
                    if (128 + tid >= p.Nk) bias = -INFINITY;
                    else if (p.mask && !p.mask[(int64_t)b * p.Nk + 128 + tid]) bias = NEG_BIG;
                    kbias[128 + tid] = bias;
                }
            }
        }
    }
    __syncthreads();
    attn_stamp(p, tid, 1);
    bf16x8 qf[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[qt][s] = rowfrag(Qs, wave * (16 * QT) + 16 * qt, s, lane);

    float m[QT], l[QT];
    f32x4 o[4][QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m[qt] = M_INIT; l[qt] = 0.f; }   // m: running max of the RAW scores
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int coff = p.Nk - p.Nq;                      // causal: query i sees keys j <= i + coff
    const int kend = p.causal ? min(p.Nk, q0 + 128 + coff) : p.Nk;
    for (int k0 = 0; k0 < kend; k0 += 128) {
        const int kval = min(128, p.Nk - k0);
        if (k0 > 0) {         // (block 0 was staged with Q)
            __syncthreads();  // previous block's K/V reads are done
            stage_tile<NT>(Ks, ksrc, p.ldk, k0, kval, p.cosT, p.sinT, p.rot, tid);
            stage_tile<NT>(Vs, vsrc, p.ldv, k0, kval, nullptr, nullptr, 0, tid);
        }
        if (tid < 128) {
            float bias = 0.f;
            if (tid >= kval) bias = -INFINITY;                                           // padding: never attended
            else if (p.mask && !p.mask[(int64_t)b * p.Nk + k0 + tid]) bias = NEG_BIG;      // masked_fill(-max)
            kbias[tid] = bias;
        }
        __syncthreads();

        // one key block: S^T tiles -> online softmax -> O^T += V^T P^T.  NKT = 16-key tiles processed: 8 for a full block, 2
        // for a short tail (the 2 keys S = 130 leaves behind would otherwise cost a whole block of MFMAs and exps)
        auto block = [&](auto nkt_c, const char* Kb, const char* Vb, const float* kbias) {
            constexpr int NKT = decltype(nkt_c)::value;
        f32x4 acc[NKT][QT];
            // S^T = K Q^T on top of the key bias (0 / masked / padding): the MFMA accumulates onto it, so masking costs no
            // vector instruction.  Scores stay RAW (unscaled) until the exponent.
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const f32x4 kb = *reinterpret_cast<const f32x4*>(kbias + 16 * kt + 4 * g);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc[kt][qt] = kb;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 kf = rowfrag(Kb, 16 * kt, s, lane);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        acc[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][s], acc[kt][qt], 0, 0, 0);
                }
            }
            mfma_results_ready();
            // causal: only a block that reaches past this wave's first query compares indices (wave-uniform test)
            if (p.causal && k0 + 16 * NKT - 1 > q0 + wave * (16 * QT) + coff) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    const int lim = q0 + wave * (16 * QT) + 16 * qt + li + coff - k0 - 4 * g;   // key offset 16kt + r allowed iff <= lim
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[kt][qt][r] = min_raw(acc[kt][qt][r], 16 * kt + r > lim ? NEG_BIG : INFINITY);
                }
            }
            // online softmax in the exp2 domain (query = lane&15 column; keys on rows 4g+reg of each key tile):
            // p = 2^((s - m) * scale * log2 e); the subtraction comes first so that equal huge values (a fully masked row)
            // give exactly 0
            float alpha[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float mx = m[qt];
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    mx = max3_raw(mx, acc[kt][qt][0], acc[kt][qt][1]);
                    mx = max3_raw(mx, acc[kt][qt][2], acc[kt][qt][3]);
                }
                const float mn = xor16_32_max(mx);
                alpha[qt] = __builtin_amdgcn_exp2f((m[qt] - mn) * SM_SCALE_LOG2E);
                m[qt] = mn;
                const f32x2 nm2 = f32x2{-mn, -mn}, c2 = f32x2{SM_SCALE_LOG2E, SM_SCALE_LOG2E};
                f32x4 ps4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const f32x2 ea = pk_mul(pk_add(lo2(acc[kt][qt]), nm2), c2);
                    const f32x2 eb = pk_mul(pk_add(hi2(acc[kt][qt]), nm2), c2);
                    // (consumers of the exponentials are compiler-visible: the trans-unit forwarding hazard is the compiler's)
                    acc[kt][qt] = f32x4{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1]),
                                        __builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
                    ps4 += acc[kt][qt];
                }
                l[qt] = l[qt] * alpha[qt] + ((ps4[0] + ps4[1]) + (ps4[2] + ps4[3]));
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt][qt] *= alpha[qt];      // (feeds the P.V MFMAs as SrcC: no inline asm)
            }
            // O^T += V^T P^T
#pragma unroll
            for (int ks = 0; ks < NKT / 2; ++ks) {
                bf16x8 pb[QT];
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) pb[qt] = pack_pair(acc[2 * ks][qt], acc[2 * ks + 1][qt]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 vf = trfrag(Vb, 32 * ks, 32 * ks + 16, 16 * dt, lane);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[qt], o[dt][qt], 0, 0, 0);
                }
            }
            };
        if (kval > 32) block(std::integral_constant<int, 8>{}, Ks, Vs, kbias);
        else block(std::integral_constant<int, 2>{}, Ks, Vs, kbias);
        if (fold_tail) {                                 // keys 128 .. Nk - 1 from their own tiles, staged with the first block
            block(std::integral_constant<int, 2>{}, Kt, Vt, kbias + 128);
            break;
        }
    }
    attn_stamp(p, tid, 2);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float lt = xor16_32_sum(l[qt]);
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        const int qi = q0 + wave * (16 * QT) + 16 * qt + li;
        if (qi < p.Nq) {
            store_row64(p.out + ((int64_t)b * p.Nq + qi) * p.ldo + h * 64, o[0][qt] * inv, o[1][qt] * inv, o[2][qt] * inv,
                        o[3][qt] * inv, g);
            if (g == 0 && p.lse) p.lse[((int64_t)b * p.H + h) * p.Nq + qi] = m[qt] * SM_SCALE + __logf(lt);
        }
    }
    if (p.stamps) {
        attn_stamp(p, tid, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        attn_stamp(p, tid, 4);
    }
}

// ================================================================================================ backward
// KV=true : owner = 128 keys of one kv head (grid: key blocks, Hkv, B); streams the queries of every head in the
//           group; writes dK (un-rotated) and dV.   S[q,key] = Q K^T ; dV^T += dO^T P ; dK^T += Q^T dS
// KV=false: owner = 128 queries of one head (grid: q blocks, H, B); streams the key blocks; writes dQ (un-rotated).
//           S^T[key,q] = K Q^T ; dQ^T += K^T dS^T
// In both, the owner's fragments sit in registers (B operand, "column" index on the lane) and the streamed tensors
// are LDS images read as row fragments (A operand) and as transposed fragments.
template <bool KV, int OT>
__global__ __launch_bounds__(128 / (16 * OT) * 64, OT == 1 ? 4 : 2) void attn_bwd_kernel(AttnParams p) {
    constexpr int NT = 128 / (16 * OT) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* R1 = smem;
    char* R2 = smem + AT_TILE;
    float* rowa = reinterpret_cast<float*>(smem + 2 * AT_TILE);  // [128] per streamed row: lse (KV) / key bias (!KV)
    float* rowb = rowa + 128;                                    // [128] per streamed row: delta (KV)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z;
    const int group = p.H / p.Hkv;
    // owner block start (dQ kernel with causal masking: heavy query blocks - the last ones - first)
    const int o0 = (!KV && p.causal ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x) * 128;
    const int hown = blockIdx.y;      // kv head (KV) or q head (!KV)
    const int hk = KV ? hown : hown / group;

    const bf16_t* kbase = p.k + (int64_t)b * p.Nk * p.ldk + p.k_off + hk * 64;
    const bf16_t* vbase = p.v + (int64_t)b * p.Nk * p.ldv + p.v_off + hk * 64;

    // The first streamed block (first query block of the group's first head / first key block) is fetched together with the
    // owner tiles - one global-load latency per workgroup instead of two.  With causal masking the dK/dV kernel starts at the
    // first query block that can see its keys.
    const int nstream = KV ? p.Nq : p.Nk;
    const int coff = p.Nk - p.Nq;
    int sfirst = 0;
    if (KV && p.causal) sfirst = ((max(o0 - coff - 127, 0) + 127) / 128) * 128;
    const bool has_first = sfirst < nstream;
    const int sval0 = min(128, nstream - sfirst);
    TileRegs<NT> f1, f2;
    // ---- owner fragments -> registers ----
    if constexpr (KV) {
        const int val = min(128, p.Nk - o0);
        TileRegs<NT> t1, t2;
        tile_load<NT>(t1, kbase, p.ldk, o0, val, p.rot, tid);
        tile_load<NT>(t2, vbase, p.ldv, o0, val, 0, tid);
        if (has_first) {
            const int hq0 = hk * group;
            tile_load<NT>(f1, p.q + (int64_t)b * p.Nq * p.ldq + p.q_off + hq0 * 64, p.ldq, sfirst, sval0, p.rot, tid);
            tile_load<NT>(f2, p.dout + (int64_t)b * p.Nq * p.ldo + hq0 * 64, p.ldo, sfirst, sval0, 0, tid);
        }
        tile_store<NT>(R1, t1, o0, val, p.cosT, p.sinT, p.rot, tid);
        tile_store<NT>(R2, t2, o0, val, nullptr, nullptr, 0, tid);
    } else {
        const int val = min(128, p.Nq - o0);
        const bf16_t* qb = p.q + (int64_t)b * p.Nq * p.ldq + p.q_off + hown * 64;
        const bf16_t* dob = p.dout + (int64_t)b * p.Nq * p.ldo + hown * 64;
        // delta[q] = sum_d dO[q][d] O[q][d] of the owner rows is computed here (4 threads per row: O from global, in flight
        // with the tile loads; dO from its LDS tile) and stored for the dK/dV kernel, which runs after this one - a
        // separate pass over dO and O (attn_delta_kernel, 34 us per layer at the bench shape) is gone
        const int drow = tid >> 2, dq4 = tid & 3;
        i32x4 o0v = i32x4{0, 0, 0, 0}, o1v = i32x4{0, 0, 0, 0};
        if (NT == 512 && drow < val) {
            const bf16_t* op = p.out + ((int64_t)b * p.Nq + o0 + drow) * p.ldo + hown * 64 + 16 * dq4;
            o0v = *reinterpret_cast<const i32x4*>(op);
            o1v = *reinterpret_cast<const i32x4*>(op + 8);
        }
        TileRegs<NT> t1, t2;
        tile_load<NT>(t1, qb, p.ldq, o0, val, p.rot, tid);
        tile_load<NT>(t2, dob, p.ldo, o0, val, 0, tid);
        if (has_first) {
            tile_load<NT>(f1, kbase, p.ldk, sfirst, sval0, p.rot, tid);
            tile_load<NT>(f2, vbase, p.ldv, sfirst, sval0, 0, tid);
        }
        tile_store<NT>(R1, t1, o0, val, p.cosT, p.sinT, p.rot, tid, p.qpos);
        tile_store<NT>(R2, t2, o0, val, nullptr, nullptr, 0, tid);
        __syncthreads();
        if (NT == 512) {
            const i32x4 d0v = *reinterpret_cast<const i32x4*>(R2 + drow * AT_STRIDE + 32 * dq4);
            const i32x4 d1v = *reinterpret_cast<const i32x4*>(R2 + drow * AT_STRIDE + 32 * dq4 + 16);
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dl += bf16lo((uint32_t)o0v[e]) * bf16lo((uint32_t)d0v[e]) + bf16hi((uint32_t)o0v[e]) * bf16hi((uint32_t)d0v[e]);
                dl += bf16lo((uint32_t)o1v[e]) * bf16lo((uint32_t)d1v[e]) + bf16hi((uint32_t)o1v[e]) * bf16hi((uint32_t)d1v[e]);
            }
            dl += __shfl_xor(dl, 1, 64);
            dl += __shfl_xor(dl, 2, 64);
            if (dq4 == 0) {
                rowb[drow] = dl;                       // (rows past the end: O and dO are zero there)
                if (drow < val) p.delta[((int64_t)b * p.H + hown) * p.Nq + o0 + drow] = dl;
            }
        }
    }
    __syncthreads();
    bf16x8 y1[OT][2], y2[OT][2];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            y1[ot][s] = rowfrag(R1, wave * (16 * OT) + 16 * ot, s, lane);
            y2[ot][s] = rowfrag(R2, wave * (16 * OT) + 16 * ot, s, lane);
        }
    // per-owner-column scalars
    // KV: ca = score bias of the owner key (0 valid / -inf masked or past the end).  !KV: ca = -lse[q] * log2 e (-inf for
    // rows past the end), cb = -delta[q] * scale: the addends of the two fused multiply-adds below
    float ca[OT], cb[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        const int oi = o0 + wave * (16 * OT) + 16 * ot + li;
        if constexpr (KV) {
            bool ok = oi < p.Nk;
            if (ok && p.mask) ok = p.mask[(int64_t)b * p.Nk + oi] != 0;
            ca[ot] = ok ? 0.f : -INFINITY;
            cb[ot] = 0.f;
        } else {
            const bool ok = oi < p.Nq;
            ca[ot] = ok ? -p.lse[((int64_t)b * p.H + hown) * p.Nq + oi] * LOG2E : -INFINITY;
            cb[ot] = ok ? -rowb[wave * (16 * OT) + 16 * ot + li] * SM_SCALE : 0.f;
        }
    }

    f32x4 g1[4][OT], g2[4][OT];  // g2: dK^T / dQ^T accumulators [d tile][owner tile]; g1: dV^T (KV only)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            g1[dt][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
            g2[dt][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    const int nheads = KV ? group : 1;
    // the prefetched first block goes to LDS as soon as the owner fragments have been read out of it: its registers are dead
    // before the loop (kept alive by a flag inside the loop they would cost 16 VGPRs in every iteration)
    if (has_first) {
        __syncthreads();
        tile_store<NT>(R1, f1, sfirst, sval0, p.cosT, p.sinT, p.rot, tid, KV ? p.qpos : 0);
        tile_store<NT>(R2, f2, sfirst, sval0, nullptr, nullptr, 0, tid);
    }
    bool first = has_first;                 // (the first block the loop reaches is block `sfirst` of head 0)
    for (int hh = 0; hh < nheads; ++hh) {
        const int hq = KV ? hk * group + hh : hown;
        for (int s0 = 0; s0 < nstream; s0 += 128) {
            if (p.causal) {   // whole streamed block on the masked side of the diagonal (uniform per workgroup)
                if (KV ? (s0 + 127 + coff < o0) : (s0 > o0 + 127 + coff)) continue;
            }
            if (!first) __syncthreads();  // previous tile fully consumed
            const int sval = min(128, nstream - s0);
            if constexpr (KV) {
                const bf16_t* qb = p.q + (int64_t)b * p.Nq * p.ldq + p.q_off + hq * 64;
                const bf16_t* dob = p.dout + (int64_t)b * p.Nq * p.ldo + hq * 64;
                if (!first) {
                    stage_tile<NT>(R1, qb, p.ldq, s0, sval, p.cosT, p.sinT, p.rot, tid, p.qpos);
                    stage_tile<NT>(R2, dob, p.ldo, s0, sval, nullptr, nullptr, 0, tid);
                }
                if (tid < 128) {
                    const bool ok = tid < sval;
                    const int64_t idx = ((int64_t)b * p.H + hq) * p.Nq + s0 + tid;
                    rowa[tid] = ok ? -p.lse[idx] * LOG2E : -INFINITY;     // P = 2^(s * scale * log2e + rowa)
                    rowb[tid] = ok ? -p.delta[idx] * SM_SCALE : 0.f;       // dS = P * (dP * scale + rowb)
                }
            } else {
                if (!first) {
                    stage_tile<NT>(R1, kbase, p.ldk, s0, sval, p.cosT, p.sinT, p.rot, tid);
                    stage_tile<NT>(R2, vbase, p.ldv, s0, sval, nullptr, nullptr, 0, tid);
                }
                if (tid < 128) {
                    bool ok = tid < sval;
                    if (ok && p.mask) ok = p.mask[(int64_t)b * p.Nk + s0 + tid] != 0;
                    rowa[tid] = ok ? 0.f : -INFINITY;                      // score bias of the streamed key
                }
            }
            first = false;
            __syncthreads();

            // waves whose owner rows are all past the end (the 2-key tail block of S = 130 keeps one wave of eight busy)
            // and streamed 32-row pairs past the end of a partial block only take part in staging and barriers
            const int nown_ = KV ? p.Nk : p.Nq;
            const bool wave_live = o0 + wave * (16 * OT) < nown_;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                if (!wave_live || 32 * pr >= sval) break;
                f32x4 sa[2][OT], dp[2][OT];  // [streamed tile of the pair][owner tile]
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int rb = 32 * pr + 16 * t;
                    // rows = streamed index 4g+r, column = owner index.  The validity bias (0 / -inf) of the key side is the
                    // initial value of the score accumulator: masking costs no vector instruction
                    const f32x4 ra = *reinterpret_cast<const f32x4*>(rowa + rb + 4 * g);
                    f32x4 rbv = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (KV) rbv = *reinterpret_cast<const f32x4*>(rowb + rb + 4 * g);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        if constexpr (KV) sa[t][ot] = f32x4{ca[ot], ca[ot], ca[ot], ca[ot]};
                        else sa[t][ot] = ra;
                        dp[t][ot] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 x1 = rowfrag(R1, rb, s, lane);
                        const bf16x8 x2 = rowfrag(R2, rb, s, lane);
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot) {
                            sa[t][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, y1[ot][s], sa[t][ot], 0, 0, 0);
                            dp[t][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x2, y2[ot][s], dp[t][ot], 0, 0, 0);
                        }
                    }
                    mfma_results_ready();
                    // P = 2^(s * scale * log2e - lse * log2e), dS = P * (dP - delta) * scale: two packed fmas, the
                    // exponentials and a packed multiply per pair of elements
                    const f32x2 c2 = f32x2{SM_SCALE_LOG2E, SM_SCALE_LOG2E}, sc2 = f32x2{SM_SCALE, SM_SCALE};
                    // causal: only tiles that touch the masked side of the diagonal compare indices
                    const int ow = o0 + wave * (16 * OT);
                    const bool diag = p.causal && (KV ? (ow + 16 * OT - 1 > s0 + rb + coff) : (s0 + rb + 15 > ow + coff));
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        f32x2 la, lb, da, db;      // addends: -lse * log2e, -delta * scale
                        if constexpr (KV) { la = lo2(ra); lb = hi2(ra); da = lo2(rbv); db = hi2(rbv); }
                        else { la = lb = f32x2{ca[ot], ca[ot]}; da = db = f32x2{cb[ot], cb[ot]}; }
                        const f32x2 ea = pk_fma(lo2(sa[t][ot]), c2, la), eb = pk_fma(hi2(sa[t][ot]), c2, lb);
                        f32x4 pv = f32x4{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1]),
                                         __builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
                        if (diag) {
                            const int oi = ow + 16 * ot + li;                            // owner index
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int si = s0 + rb + 4 * g + r;                      // streamed index
                                if (KV ? (oi > si + coff) : (si > oi + coff)) pv[r] = 0.f;   // key > query
                            }
                        }
                        const f32x2 ta = pk_fma(lo2(dp[t][ot]), sc2, da), tb = pk_fma(hi2(dp[t][ot]), sc2, db);
                        sa[t][ot] = pv;
                        dp[t][ot] = pv * f32x4{ta[0], ta[1], tb[0], tb[1]};   // (reads exponentials: compiler-visible multiply)
                    }
                }
                bf16x8 pb[OT], dsb[OT];
#pragma unroll
                for (int ot = 0; ot < OT; ++ot) {
                    pb[ot] = pack_pair(sa[0][ot], sa[1][ot]);
                    dsb[ot] = pack_pair(dp[0][ot], dp[1][ot]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 a1 = trfrag(R1, 32 * pr, 32 * pr + 16, 16 * dt, lane);
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot)
                        g2[dt][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, dsb[ot], g2[dt][ot], 0, 0, 0);
                    if constexpr (KV) {
                        const bf16x8 a2 = trfrag(R2, 32 * pr, 32 * pr + 16, 16 * dt, lane);
#pragma unroll
                        for (int ot = 0; ot < OT; ++ot)
                            g1[dt][ot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, pb[ot], g1[dt][ot], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: un-rotate (transpose of the rotary map) and store; lane holds d = 16dt + 4g + r for its column
    const int nown = KV ? p.Nk : p.Nq;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        const int oi = o0 + wave * (16 * OT) + 16 * ot + li;
        if (oi >= nown) continue;
        if (p.rot) {
            const int hts = p.rot >> 5;                 // 16-dim tiles per rotary half (1 or 2)
#pragma unroll
            for (int ht = 0; ht < 2; ++ht) {
                if (ht >= hts) break;
                const int64_t pos = oi + (KV ? 0 : p.qpos);
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.cosT + pos * (p.rot >> 1) + 16 * ht + 4 * g);
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.sinT + pos * (p.rot >> 1) + 16 * ht + 4 * g);
                f32x4& lo = hts == 1 ? g2[0][ot] : g2[ht][ot];
                f32x4& hi = hts == 1 ? g2[1][ot] : g2[ht + 2][ot];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = lo[r], bb = hi[r];
                    lo[r] = a * c4[r] + bb * s4[r];
                    hi[r] = bb * c4[r] - a * s4[r];
                }
            }
        }
        if constexpr (KV) {
            bf16_t* dkp = p.dk + ((int64_t)b * p.Nk + oi) * p.ldk + p.k_off + hk * 64 + 4 * g;
            bf16_t* dvp = p.dv + ((int64_t)b * p.Nk + oi) * p.ldv + p.v_off + hk * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                i32x2 w;
                w[0] = (int)pack_bf16x2(g2[dt][ot][0], g2[dt][ot][1]);
                w[1] = (int)pack_bf16x2(g2[dt][ot][2], g2[dt][ot][3]);
                *reinterpret_cast<i32x2*>(dkp + 16 * dt) = w;
                w[0] = (int)pack_bf16x2(g1[dt][ot][0], g1[dt][ot][1]);
                w[1] = (int)pack_bf16x2(g1[dt][ot][2], g1[dt][ot][3]);
                *reinterpret_cast<i32x2*>(dvp + 16 * dt) = w;
            }
        } else {
            bf16_t* dqp = p.dq + ((int64_t)b * p.Nq + oi) * p.ldq + p.q_off + hown * 64 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                i32x2 w;
                w[0] = (int)pack_bf16x2(g2[dt][ot][0], g2[dt][ot][1]);
                w[1] = (int)pack_bf16x2(g2[dt][ot][2], g2[dt][ot][3]);
                *reinterpret_cast<i32x2*>(dqp + 16 * dt) = w;
            }
        }
    }
}

// ---- fused backward for one block of queries AND keys (the DiT's self-attention: 126 tokens) ----------------------------------
// At this size the two-kernel backward is bound by HBM, not by MFMA: each kernel re-reads q, k, v and dO.  Here one workgroup per
// (batch, head) stages all four tiles once (4 x 20 KiB = exactly half a CU's LDS, so two workgroups share a CU and one's staging
// runs under the other's MFMAs; the per-row softmax statistics live in the 32 pad bytes of the tile rows) and runs both
// orientations back to back on them: keys as owners -> dK, dV (queries streamed from LDS), then queries as owners -> dQ (keys
// streamed from LDS).  No barrier between the two: LDS is read-only after staging.  Same arithmetic per element as the two
// kernels (same operand rounding, same sum order per accumulator).
__device__ __forceinline__ float& row_stat(char* tile, int row, int slot) {
    return *reinterpret_cast<float*>(tile + row * AT_STRIDE + 128 + 4 * slot);
}

__global__ __launch_bounds__(512, 4) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_bwd_fused_kernel(AttnParams p) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* RQ = smem;                  // pad slot 0: -lse * log2 e (-inf past the end), slot 1: -delta * scale
    char* RD = smem + AT_TILE;        // dO
    char* RK = smem + 2 * AT_TILE;    // pad slot 0: key bias (0 valid / -inf masked or past the end)
    char* RV = smem + 3 * AT_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int N = p.Nq, Nk = p.Nk;
    const bf16_t* qb = p.q + (int64_t)b * N * p.ldq + p.q_off + h * 64;
    const bf16_t* kb = p.k + (int64_t)b * Nk * p.ldk + p.k_off + h * 64;
    const bf16_t* vb = p.v + (int64_t)b * Nk * p.ldv + p.v_off + h * 64;
    const bf16_t* dob = p.dout + (int64_t)b * N * p.ldo + h * 64;
    attn_stamp(p, tid, 0);

    // ---- staging: all global loads in flight together, then rotary + LDS writes
    {
        TileRegs<NT> tq, td, tk, tv;
        tile_load<NT>(tq, qb, p.ldq, 0, N, p.rot, tid);
        tile_load<NT>(td, dob, p.ldo, 0, N, 0, tid);
        tile_load<NT>(tk, kb, p.ldk, 0, Nk, p.rot, tid);
        tile_load<NT>(tv, vb, p.ldv, 0, Nk, 0, tid);
        const RopeRegs rq = rope_load(p.cosT, p.sinT, 0, N, p.rot, tid, p.qpos);
        const RopeRegs rk = rope_load(p.cosT, p.sinT, 0, Nk, p.rot, tid);
        // delta[q] = sum_d dO[q][d] O[q][d]: 4 threads per row, O straight from global (in flight with the tiles)
        const int drow = tid >> 2, dq4 = tid & 3;
        i32x4 o0v = i32x4{0, 0, 0, 0}, o1v = i32x4{0, 0, 0, 0};
        if (drow < N) {
            const bf16_t* op = p.out + ((int64_t)b * N + drow) * p.ldo + h * 64 + 16 * dq4;
            o0v = *reinterpret_cast<const i32x4*>(op);
            o1v = *reinterpret_cast<const i32x4*>(op + 8);
        }
        // the row statistics' inputs too (log-sum-exp of the query, mask byte of the key): requested now, used after the barrier
        float lse_pre = 0.f;
        bool key_ok = drow < Nk;
        if (dq4 == 0 && drow < N) lse_pre = p.lse[((int64_t)b * p.H + h) * N + drow];
        if (dq4 == 1 && key_ok && p.mask) key_ok = p.mask[(int64_t)b * Nk + drow] != 0;
        tile_store<NT>(RQ, tq, 0, N, p.cosT, p.sinT, p.rot, tid, p.qpos, p.rot != 0, rq);
        tile_store<NT>(RD, td, 0, N, nullptr, nullptr, 0, tid);
        tile_store<NT>(RK, tk, 0, Nk, p.cosT, p.sinT, p.rot, tid, 0, p.rot != 0, rk);
        tile_store<NT>(RV, tv, 0, Nk, nullptr, nullptr, 0, tid);
        __syncthreads();
        attn_stamp(p, tid, 1);
        const i32x4 d0v = *reinterpret_cast<const i32x4*>(RD + drow * AT_STRIDE + 32 * dq4);
        const i32x4 d1v = *reinterpret_cast<const i32x4*>(RD + drow * AT_STRIDE + 32 * dq4 + 16);
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dl += bf16lo((uint32_t)o0v[e]) * bf16lo((uint32_t)d0v[e]) + bf16hi((uint32_t)o0v[e]) * bf16hi((uint32_t)d0v[e]);
            dl += bf16lo((uint32_t)o1v[e]) * bf16lo((uint32_t)d1v[e]) + bf16hi((uint32_t)o1v[e]) * bf16hi((uint32_t)d1v[e]);
        }
        dl += __shfl_xor(dl, 1, 64);
        dl += __shfl_xor(dl, 2, 64);
        if (dq4 == 0) {
            const bool ok = drow < N;
            row_stat(RQ, drow, 0) = ok ? -lse_pre * LOG2E : -INFINITY;
            row_stat(RQ, drow, 1) = ok ? -dl * SM_SCALE : 0.f;
            if (ok) p.delta[((int64_t)b * p.H + h) * N + drow] = dl;      // (kept for callers that look at it)
        } else if (dq4 == 1) {
            row_stat(RK, drow, 0) = key_ok ? 0.f : -INFINITY;
        }
        __syncthreads();
    }
    attn_stamp(p, tid, 2);
    const f32x2 c2 = f32x2{SM_SCALE_LOG2E, SM_SCALE_LOG2E}, sc2 = f32x2{SM_SCALE, SM_SCALE};
    const int orow = wave * 16;                  // this wave's 16 owner rows (keys in phase A, queries in phase B)

    // =============================== phase A: keys own, queries stream -> dK, dV
    if (orow < Nk) {
        bf16x8 y1[2], y2[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) { y1[s] = rowfrag(RK, orow, s, lane); y2[s] = rowfrag(RV, orow, s, lane); }
        const float ca = row_stat(RK, orow + li, 0);
        f32x4 g1[4], g2[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { g1[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; g2[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if (32 * pr >= N) break;
            f32x4 sa[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rb = 32 * pr + 16 * t;
                sa[t] = f32x4{ca, ca, ca, ca};
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(RQ, rb, s, lane), y1[s], sa[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(RD, rb, s, lane), y2[s], dp[t], 0, 0, 0);
                }
                f32x4 ra, rbv;          // streamed queries rb + 4g + r: -lse * log2 e, -delta * scale
#pragma unroll
                for (int r = 0; r < 4; ++r) { ra[r] = row_stat(RQ, rb + 4 * g + r, 0); rbv[r] = row_stat(RQ, rb + 4 * g + r, 1); }
                mfma_results_ready();
                const f32x2 ea = pk_fma(lo2(sa[t]), c2, lo2(ra)), eb = pk_fma(hi2(sa[t]), c2, hi2(ra));
                const f32x4 pv = f32x4{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1]),
                                       __builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
                const f32x2 ta = pk_fma(lo2(dp[t]), sc2, lo2(rbv)), tb = pk_fma(hi2(dp[t]), sc2, hi2(rbv));
                sa[t] = pv;
                dp[t] = pv * f32x4{ta[0], ta[1], tb[0], tb[1]};
            }
            const bf16x8 pb = pack_pair(sa[0], sa[1]), dsb = pack_pair(dp[0], dp[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                g2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag(RQ, 32 * pr, 32 * pr + 16, 16 * dt, lane), dsb, g2[dt], 0, 0, 0);
                g1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag(RD, 32 * pr, 32 * pr + 16, 16 * dt, lane), pb, g1[dt], 0, 0, 0);
            }
        }
        attn_stamp(p, tid, 3);
        const int oi = orow + li;
        if (oi < Nk) {
            if (p.rot) {
                const int hts = p.rot >> 5;
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    if (ht >= hts) break;
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.cosT + (int64_t)oi * (p.rot >> 1) + 16 * ht + 4 * g);
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.sinT + (int64_t)oi * (p.rot >> 1) + 16 * ht + 4 * g);
                    f32x4& lo = hts == 1 ? g2[0] : g2[ht];
                    f32x4& hi = hts == 1 ? g2[1] : g2[ht + 2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = lo[r], bb = hi[r];
                        lo[r] = a * c4[r] + bb * s4[r];
                        hi[r] = bb * c4[r] - a * s4[r];
                    }
                }
            }
            store_row64(p.dk + ((int64_t)b * Nk + oi) * p.ldk + p.k_off + h * 64, g2[0], g2[1], g2[2], g2[3], g);
            store_row64(p.dv + ((int64_t)b * Nk + oi) * p.ldv + p.v_off + h * 64, g1[0], g1[1], g1[2], g1[3], g);
        }
    }

    // =============================== phase B: queries own, keys stream -> dQ
    attn_stamp(p, tid, 4);
    if (orow < N) {
        bf16x8 y1[2], y2[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) { y1[s] = rowfrag(RQ, orow, s, lane); y2[s] = rowfrag(RD, orow, s, lane); }
        const float ca = row_stat(RQ, orow + li, 0), cb = row_stat(RQ, orow + li, 1);
        f32x4 g2[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) g2[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if (32 * pr >= Nk) break;
            f32x4 sa[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rb = 32 * pr + 16 * t;
#pragma unroll
                for (int r = 0; r < 4; ++r) sa[t][r] = row_stat(RK, rb + 4 * g + r, 0);     // key bias of the streamed keys
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(RK, rb, s, lane), y1[s], sa[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(RV, rb, s, lane), y2[s], dp[t], 0, 0, 0);
                }
                mfma_results_ready();
                const f32x2 la = f32x2{ca, ca}, da = f32x2{cb, cb};
                const f32x2 ea = pk_fma(lo2(sa[t]), c2, la), eb = pk_fma(hi2(sa[t]), c2, la);
                const f32x4 pv = f32x4{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1]),
                                       __builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
                const f32x2 ta = pk_fma(lo2(dp[t]), sc2, da), tb = pk_fma(hi2(dp[t]), sc2, da);
                dp[t] = pv * f32x4{ta[0], ta[1], tb[0], tb[1]};
            }
            const bf16x8 dsb = pack_pair(dp[0], dp[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                g2[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag(RK, 32 * pr, 32 * pr + 16, 16 * dt, lane), dsb, g2[dt], 0, 0, 0);
        }
        const int oi = orow + li;
        if (oi < N) {
            if (p.rot) {
                const int hts = p.rot >> 5;
#pragma unroll
                for (int ht = 0; ht < 2; ++ht) {
                    if (ht >= hts) break;
                    const int64_t pos = oi + p.qpos;
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(p.cosT + pos * (p.rot >> 1) + 16 * ht + 4 * g);
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.sinT + pos * (p.rot >> 1) + 16 * ht + 4 * g);
                    f32x4& lo = hts == 1 ? g2[0] : g2[ht];
                    f32x4& hi = hts == 1 ? g2[1] : g2[ht + 2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = lo[r], bb = hi[r];
                        lo[r] = a * c4[r] + bb * s4[r];
                        hi[r] = bb * c4[r] - a * s4[r];
                    }
                }
            }
            attn_stamp(p, tid, 5);
            store_row64(p.dq + ((int64_t)b * N + oi) * p.ldq + p.q_off + h * 64, g2[0], g2[1], g2[2], g2[3], g);
        }
    }
    if (p.stamps) {
        attn_stamp(p, tid, 6);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        attn_stamp(p, tid, 7);
    }
}

// ---- the same for the DiT's cross-attention: `group` query heads per kv head and a few keys beyond the 128-row block --------------
// One workgroup per (batch, kv head).  Pass-major, so that no accumulator outlives its pass (128 VGPRs keep two workgroups on a CU):
//   pass A: for every head of the group stage Q, dO and accumulate dK, dV of keys 0..127; store them;
//   pass B: for every head (last-staged first) dQ over all keys; store it.
// Keys 128 .. Nk-1 (the 130-token context of the bench has two) ride in the rows of the Q / dO tiles that the <= 126 queries leave
// free: as a ninth streamed key tile in pass B (rows 112..127 of the Q / dO images, the queries among them masked by a key bias
// of -inf) and, for their own dK / dV, as a ninth owner tile that wave 7 walks after its pass-B work.  No rotary here (the
// cross-attention has none); K / V of a head group are read once, Q / dO of the first head twice (second time from L2).
__global__ __launch_bounds__(512, 4) void attn_bwd_fused_gqa_kernel(AttnParams p) {
    constexpr int NT = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* RQ = smem;                  // pad slots: 0 = -lse * log2 e (query role), 1 = -delta * scale, 2 = key bias (key role, rows >= 112)
    char* RD = smem + AT_TILE;
    char* RK = smem + 2 * AT_TILE;    // pad slot 0: key bias of keys 0..127
    char* RV = smem + 3 * AT_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int N = p.Nq, Nk = p.Nk;
    const int nmain = min(Nk, 128), tail = Nk - nmain;        // tail keys live in rows [128 - tail, 128) of RQ / RD
    const int trow0 = 128 - tail;
    const bf16_t* kb = p.k + (int64_t)b * Nk * p.ldk + p.k_off + hk * 64;
    const bf16_t* vb = p.v + (int64_t)b * Nk * p.ldv + p.v_off + hk * 64;
    const f32x2 c2 = f32x2{SM_SCALE_LOG2E, SM_SCALE_LOG2E}, sc2 = f32x2{SM_SCALE, SM_SCALE};
    const int orow = wave * 16;
    const int drow = tid >> 2, dq4 = tid & 3;

    // ---- stage Q, dO of head h (+ the tail keys into the free rows, + per-row statistics); K, V with the first head
    auto stage_head = [&](int h, bool with_kv, bool write_delta) {
        const bf16_t* qb = p.q + (int64_t)b * N * p.ldq + p.q_off + h * 64;
        const bf16_t* dob = p.dout + (int64_t)b * N * p.ldo + h * 64;
        TileRegs<NT> tq, td, tk, tv;
        tile_load<NT>(tq, qb, p.ldq, 0, N, 0, tid);
        tile_load<NT>(td, dob, p.ldo, 0, N, 0, tid);
        if (with_kv) {
            tile_load<NT>(tk, kb, p.ldk, 0, nmain, 0, tid);
            tile_load<NT>(tv, vb, p.ldv, 0, nmain, 0, tid);
        }
        i32x4 o0v = i32x4{0, 0, 0, 0}, o1v = i32x4{0, 0, 0, 0};
        if (drow < N) {
            const bf16_t* op = p.out + ((int64_t)b * N + drow) * p.ldo + h * 64 + 16 * dq4;
            o0v = *reinterpret_cast<const i32x4*>(op);
            o1v = *reinterpret_cast<const i32x4*>(op + 8);
        }
        i32x4 tkv = i32x4{0, 0, 0, 0}, tvv = i32x4{0, 0, 0, 0};      // tail keys: 8 threads x 16 bytes per row
        const int tj = tid >> 3, tc = tid & 7;
        if (tj < tail) {
            tkv = *reinterpret_cast<const i32x4*>(kb + (int64_t)(128 + tj) * p.ldk + 8 * tc);
            tvv = *reinterpret_cast<const i32x4*>(vb + (int64_t)(128 + tj) * p.ldv + 8 * tc);
        }
        // inputs of the per-row statistics, requested with the tiles (behind the barrier each was one more exposed latency)
        float lse_pre = 0.f;
        bool stat_ok = dq4 == 1 ? drow >= trow0 : drow < nmain;
        if (dq4 == 0 && drow < N) lse_pre = p.lse[((int64_t)b * p.H + h) * N + drow];
        if (dq4 == 1 && stat_ok && p.mask) stat_ok = p.mask[(int64_t)b * Nk + 128 + (drow - trow0)] != 0;
        if (dq4 == 2 && with_kv && stat_ok && p.mask) stat_ok = p.mask[(int64_t)b * Nk + drow] != 0;
        tile_store<NT>(RQ, tq, 0, N, nullptr, nullptr, 0, tid);
        tile_store<NT>(RD, td, 0, N, nullptr, nullptr, 0, tid);
        if (with_kv) {
            tile_store<NT>(RK, tk, 0, nmain, nullptr, nullptr, 0, tid);
            tile_store<NT>(RV, tv, 0, nmain, nullptr, nullptr, 0, tid);
        }
        __syncthreads();
        if (tj < tail) {
            *reinterpret_cast<i32x4*>(RQ + (trow0 + tj) * AT_STRIDE + 16 * tc) = tkv;
            *reinterpret_cast<i32x4*>(RD + (trow0 + tj) * AT_STRIDE + 16 * tc) = tvv;
        }
        const i32x4 d0v = *reinterpret_cast<const i32x4*>(RD + drow * AT_STRIDE + 32 * dq4);
        const i32x4 d1v = *reinterpret_cast<const i32x4*>(RD + drow * AT_STRIDE + 32 * dq4 + 16);
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dl += bf16lo((uint32_t)o0v[e]) * bf16lo((uint32_t)d0v[e]) + bf16hi((uint32_t)o0v[e]) * bf16hi((uint32_t)d0v[e]);
            dl += bf16lo((uint32_t)o1v[e]) * bf16lo((uint32_t)d1v[e]) + bf16hi((uint32_t)o1v[e]) * bf16hi((uint32_t)d1v[e]);
        }
        dl += __shfl_xor(dl, 1, 64);
        dl += __shfl_xor(dl, 2, 64);
        if (dq4 == 0) {
            const bool ok = drow < N;
            row_stat(RQ, drow, 0) = ok ? -lse_pre * LOG2E : -INFINITY;
            row_stat(RQ, drow, 1) = ok ? -dl * SM_SCALE : 0.f;
            if (ok && write_delta) p.delta[((int64_t)b * p.H + h) * N + drow] = dl;
        } else if (dq4 == 1) {
            row_stat(RQ, drow, 2) = stat_ok ? 0.f : -INFINITY;          // this row holds tail key 128 + (drow - trow0)
        } else if (dq4 == 2 && with_kv) {
            row_stat(RK, drow, 0) = stat_ok ? 0.f : -INFINITY;
        }
        __syncthreads();
    };

    // ---- keys own (fragments y1 / y2 of 16 key rows, bias ca), the queries of the staged head stream: dK^T, dV^T +=
    auto keys_own = [&](const bf16x8 (&y1)[2], const bf16x8 (&y2)[2], float ca, f32x4 (&gv)[4], f32x4 (&gk)[4]) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if (32 * pr >= N) break;
            f32x4 sa[2], dp[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rb = 32 * pr + 16 * t;
                sa[t] = f32x4{ca, ca, ca, ca};
                dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    sa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(RQ, rb, s, lane), y1[s], sa[t], 0, 0, 0);
                    dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(RD, rb, s, lane), y2[s], dp[t], 0, 0, 0);
                }
                f32x4 ra, rbv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { ra[r] = row_stat(RQ, rb + 4 * g + r, 0); rbv[r] = row_stat(RQ, rb + 4 * g + r, 1); }
                mfma_results_ready();
                const f32x2 ea = pk_fma(lo2(sa[t]), c2, lo2(ra)), eb = pk_fma(hi2(sa[t]), c2, hi2(ra));
                f32x4 pv = f32x4{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1]),
                                 __builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
                const f32x2 ta = pk_fma(lo2(dp[t]), sc2, lo2(rbv)), tb = pk_fma(hi2(dp[t]), sc2, hi2(rbv));
                f32x4 ds = pv * f32x4{ta[0], ta[1], tb[0], tb[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {            // rows past the queries hold tail keys, not dO: their dP is not a number to keep
                    if (rb + 4 * g + r >= N) { pv[r] = 0.f; ds[r] = 0.f; }
                }
                sa[t] = pv;
                dp[t] = ds;
            }
            const bf16x8 pb = pack_pair(sa[0], sa[1]), dsb = pack_pair(dp[0], dp[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                gk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag(RQ, 32 * pr, 32 * pr + 16, 16 * dt, lane), dsb, gk[dt], 0, 0, 0);
                gv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag(RD, 32 * pr, 32 * pr + 16, 16 * dt, lane), pb, gv[dt], 0, 0, 0);
            }
        }
    };

    // ================================= pass A: dK, dV of keys 0..127, summed over the heads of the group
    {
        stage_head(hk * group, true, true);       // (before the accumulators exist: its 48 registers of loads in flight are not theirs)
        f32x4 g1[4], g2[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { g1[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; g2[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int hh = 0; hh < group; ++hh) {
            if (hh > 0) {
                __syncthreads();
                stage_head(hk * group + hh, false, true);
            }
            if (orow < nmain) {
                bf16x8 y1[2], y2[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) { y1[s] = rowfrag(RK, orow, s, lane); y2[s] = rowfrag(RV, orow, s, lane); }
                keys_own(y1, y2, row_stat(RK, orow + li, 0), g1, g2);
            }
        }
        const int oi = orow + li;
        if (oi < nmain) {
            store_row64(p.dk + ((int64_t)b * Nk + oi) * p.ldk + p.k_off + hk * 64, g2[0], g2[1], g2[2], g2[3], g);
            store_row64(p.dv + ((int64_t)b * Nk + oi) * p.ldv + p.v_off + hk * 64, g1[0], g1[1], g1[2], g1[3], g);
        }
    }

    // ================================= pass B: dQ of every head (the last-staged one first); wave 7 also gathers dK, dV of the tail keys
    for (int hh = group - 1; hh >= 0; --hh) {
        const int h = hk * group + hh;
        if (hh != group - 1) {
            __syncthreads();
            stage_head(h, false, false);
        }
        if (orow < N) {
            bf16x8 y1[2], y2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) { y1[s] = rowfrag(RQ, orow, s, lane); y2[s] = rowfrag(RD, orow, s, lane); }
            const float ca = row_stat(RQ, orow + li, 0), cb = row_stat(RQ, orow + li, 1);
            f32x4 gq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) gq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // streamed key tiles: pairs 0..3 from RK / RV, then (tail keys) rows 112..127 of RQ / RD paired with nothing
            auto stream_keys = [&](const char* SK, const char* SV, int base, auto ext_c) {
                constexpr bool EXT = decltype(ext_c)::value;
                f32x4 dp[2];
#pragma unroll
                for (int t = 0; t < (EXT ? 1 : 2); ++t) {
                    dp[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const int rb = base + 16 * t;
                    f32x4 sa;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sa[r] = EXT ? row_stat(RQ, rb + 4 * g + r, 2) : row_stat(RK, rb + 4 * g + r, 0);
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(SK, rb, s, lane), y1[s], sa, 0, 0, 0);
                        dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rowfrag(SV, rb, s, lane), y2[s], dp[t], 0, 0, 0);
                    }
                    mfma_results_ready();
                    const f32x2 la = f32x2{ca, ca}, da = f32x2{cb, cb};
                    const f32x2 ea = pk_fma(lo2(sa), c2, la), eb = pk_fma(hi2(sa), c2, la);
                    const f32x4 pv = f32x4{__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1]),
                                           __builtin_amdgcn_exp2f(eb[0]), __builtin_amdgcn_exp2f(eb[1])};
                    const f32x2 ta = pk_fma(lo2(dp[t]), sc2, da), tb = pk_fma(hi2(dp[t]), sc2, da);
                    f32x4 ds = pv * f32x4{ta[0], ta[1], tb[0], tb[1]};
                    if constexpr (EXT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (rb + 4 * g + r < trow0) ds[r] = 0.f;      // a query row in the key role: P is 0, dP is not a number to keep
                    }
                    dp[t] = ds;
                }
                if constexpr (EXT) dp[1] = f32x4{0.f, 0.f, 0.f, 0.f};
                const bf16x8 dsb = pack_pair(dp[0], dp[1]);
                const int rb_ = EXT ? base : base + 16;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    gq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(trfrag(SK, base, rb_, 16 * dt, lane), dsb, gq[dt], 0, 0, 0);
            };
#pragma unroll 1
            for (int pr = 0; pr < 4; ++pr) {
                if (32 * pr >= nmain) break;
                stream_keys(RK, RV, 32 * pr, std::false_type{});
            }
            if (tail > 0) stream_keys(RQ, RD, 112, std::true_type{});
            const int oi = orow + li;
            if (oi < N) {
                store_row64(p.dq + ((int64_t)b * N + oi) * p.ldq + p.q_off + h * 64, gq[0], gq[1], gq[2], gq[3], g);
            }
        }
        if (tail > 0 && wave == 7) {        // the ninth owner tile: rows 112..127 of the Q / dO images in the key role
            // its accumulators rest between heads in the pad bytes of the dO / V tile rows 2 * lane, 2 * lane + 1 (staging writes only
            // the first 128 bytes of a row): held in registers across the dQ work above they cost 20 spilled VGPRs
            f32x4 gt1[4], gt2[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                if (hh == group - 1) { gt1[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; gt2[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                else {
                    gt1[dt] = *reinterpret_cast<const f32x4*>(RD + (2 * lane + (dt >> 1)) * AT_STRIDE + 128 + 16 * (dt & 1));
                    gt2[dt] = *reinterpret_cast<const f32x4*>(RV + (2 * lane + (dt >> 1)) * AT_STRIDE + 128 + 16 * (dt & 1));
                }
            }
            bf16x8 y1[2], y2[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) { y1[s] = rowfrag(RQ, 112, s, lane); y2[s] = rowfrag(RD, 112, s, lane); }
            keys_own(y1, y2, row_stat(RQ, 112 + li, 2), gt1, gt2);
            if (hh > 0) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<f32x4*>(RD + (2 * lane + (dt >> 1)) * AT_STRIDE + 128 + 16 * (dt & 1)) = gt1[dt];
                    *reinterpret_cast<f32x4*>(RV + (2 * lane + (dt >> 1)) * AT_STRIDE + 128 + 16 * (dt & 1)) = gt2[dt];
                }
            } else {
                const int row = 112 + li;
                if (row >= trow0) {
                    const int key = 128 + (row - trow0);
                    store_row64(p.dk + ((int64_t)b * Nk + key) * p.ldk + p.k_off + hk * 64, gt2[0], gt2[1], gt2[2], gt2[3], g);
                    store_row64(p.dv + ((int64_t)b * Nk + key) * p.ldv + p.v_off + hk * 64, gt1[0], gt1[1], gt1[2], gt1[3], g);
                }
            }
        }
    }
}

bool check_common(const void* q, int64_t ldq, int q_off, const void* k, int64_t ldk, int k_off, const void* v,
                  int64_t ldv, int v_off, int64_t ldo, int rot, int B, int H, int Hkv, int Nq, int Nk) {
    if (!q || !k || !v || B <= 0 || H <= 0 || Hkv <= 0 || Nq <= 0 || Nk <= 0) return false;
    if (H % Hkv) return false;
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (q_off & 7) || (k_off & 7) || (v_off & 7)) return false;
    if (rot != 0 && rot != 32 && rot != 64) return false;
    if (H > 65535 || B > 65535) return false;
    return true;
}

// ---- single-query attention: decoding against a KV cache (Nq == 1) -------------------------------------------
// The tiled kernel would spend a 128-query MFMA tile on one query.  Here a workgroup owns one (head, batch): every
// thread scores one key at a time on the vector ALUs (64-dim dot against the rotated query kept in LDS, rotary applied
// to the key from its row index), block softmax over the scores in LDS, then P.V with 8 threads per key row (16-byte
// loads of v) and a shuffle + LDS reduction over the 32 key groups.  Operands are rounded to bf16 at the same points as
// the tiled kernel (rotated q / k, probabilities) so both paths agree to rounding of the fp32 sums.
template <int ROT>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sc = reinterpret_cast<float*>(smem);  // [Nk] scores, then probabilities
    __shared__ float qs[64];
    __shared__ float red[8];
    __shared__ float osum[4][64];
    constexpr int HALF = ROT / 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const bf16_t* qsrc = p.q + (int64_t)b * p.ldq + p.q_off + h * 64;
    const bf16_t* ksrc = p.k + (int64_t)b * p.Nk * p.ldk + p.k_off + hk * 64;
    const bf16_t* vsrc = p.v + (int64_t)b * p.Nk * p.ldv + p.v_off + hk * 64;
    auto rnd = [](float x) { return bf16_to_f32(f32_to_bf16(x)); };

    if (tid < 64) {
        float x = bf16_to_f32(qsrc[tid]);
        if constexpr (ROT > 0) {
            if (tid < ROT) {
                const bool lo = tid < HALF;
                const float partner = bf16_to_f32(qsrc[lo ? tid + HALF : tid - HALF]);
                const int i = lo ? tid : tid - HALF;
                const float c = p.cosT[(int64_t)p.qpos * HALF + i], sn = p.sinT[(int64_t)p.qpos * HALF + i];
                x = rnd(x * c + (lo ? -partner : partner) * sn);
            }
        }
        qs[tid] = x;
    }
    __syncthreads();

    float mx = NEG_BIG;
    for (int j = tid; j < p.Nk; j += 256) {
        const bf16_t* kp = ksrc + (int64_t)j * p.ldk;
        float kf[64];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const i32x4 w = reinterpret_cast<const i32x4*>(kp)[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kf[8 * c + 2 * e] = bf16lo((uint32_t)w[e]);
                kf[8 * c + 2 * e + 1] = bf16hi((uint32_t)w[e]);
            }
        }
        if constexpr (ROT > 0) {
            const float* cp = p.cosT + (int64_t)j * HALF;
            const float* sp = p.sinT + (int64_t)j * HALF;
#pragma unroll
            for (int i4 = 0; i4 < HALF / 4; ++i4) {
                const f32x4 c = reinterpret_cast<const f32x4*>(cp)[i4], sn = reinterpret_cast<const f32x4*>(sp)[i4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * i4 + e;
                    const float a = kf[i], bb = kf[i + HALF];
                    kf[i] = rnd(a * c[e] - bb * sn[e]);
                    kf[i + HALF] = rnd(bb * c[e] + a * sn[e]);
                }
            }
        }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int d = 0; d < 64; d += 2) { a0 += kf[d] * qs[d]; a1 += kf[d + 1] * qs[d + 1]; }
        float sv = (a0 + a1) * SM_SCALE;
        if (p.mask && !p.mask[(int64_t)b * p.Nk + j]) sv = NEG_BIG;
        sc[j] = sv;
        mx = fmaxf(mx, sv);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float ls = 0.f;
    for (int j = tid; j < p.Nk; j += 256) {
        const float pv = __expf(sc[j] - m);
        ls += pv;
        sc[j] = rnd(pv);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, 64);
    if (lane == 0) red[4 + wave] = ls;
    __syncthreads();
    const float l = red[4] + red[5] + red[6] + red[7];

    const int kg = tid >> 3, d8 = tid & 7;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 4
    for (int j = kg; j < p.Nk; j += 32) {
        const i32x4 w = *reinterpret_cast<const i32x4*>(vsrc + (int64_t)j * p.ldv + 8 * d8);
        const float pj = sc[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[2 * e] += pj * bf16lo((uint32_t)w[e]);
            acc[2 * e + 1] += pj * bf16hi((uint32_t)w[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        acc[e] += __shfl_xor(acc[e], 8, 64);
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
    }
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) osum[wave][8 * lane + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        const float o = (osum[0][tid] + osum[1][tid] + osum[2][tid] + osum[3][tid]) / l;
        p.out[(int64_t)b * p.ldo + h * 64 + tid] = f32_to_bf16(o);
        if (tid == 0 && p.lse) p.lse[(int64_t)b * p.H + h] = m + __logf(l);
    }
}

}  // namespace

static unsigned long long* g_attn_stamps = nullptr;      // diagnostics only
extern "C" int kalle_attn_debug_stamps(void* buf) { g_attn_stamps = static_cast<unsigned long long*>(buf); return KALLE_OK; }

extern "C" int kalle_attention_fwd(const void* q, int64_t ldq, int q_off, const void* k, int64_t ldk, int k_off,
                                   const void* v, int64_t ldv, int v_off, void* out, int64_t ldo, float* lse,
                                   const float* rope_cos, const float* rope_sin, int rot, const uint8_t* key_mask,
                                   int causal, int B, int H, int Hkv, int Nq, int Nk, void* stream) {
    if (!out || !check_common(q, ldq, q_off, k, ldk, k_off, v, ldv, v_off, ldo, rot, B, H, Hkv, Nq, Nk))
        return KALLE_ERR_ARG;
    if (rot && (!rope_cos || !rope_sin)) return KALLE_ERR_ARG;
    AttnParams p{};
    p.q = static_cast<const bf16_t*>(q); p.ldq = ldq; p.q_off = q_off;
    p.k = static_cast<const bf16_t*>(k); p.ldk = ldk; p.k_off = k_off;
    p.v = static_cast<const bf16_t*>(v); p.ldv = ldv; p.v_off = v_off;
    p.out = static_cast<bf16_t*>(out); p.ldo = ldo; p.lse = lse;
    p.cosT = rope_cos; p.sinT = rope_sin; p.rot = rot; p.mask = key_mask;
    p.B = B; p.H = H; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.causal = causal;
    p.stamps = g_attn_stamps;
    if (causal && Nk < Nq) return KALLE_ERR_ARG;
    p.qpos = causal ? Nk - Nq : 0;
    if (Nq == 1 && Nk <= 15360) {   // decoding against a KV cache: scores of all keys fit in LDS (60 KB)
        const dim3 grid(H, B), block(256);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (rot == 64) KALLE_LAUNCH(attn_decode_kernel<64>, grid, block, (size_t)Nk * 4, st, p);
        else if (rot == 32) KALLE_LAUNCH(attn_decode_kernel<32>, grid, block, (size_t)Nk * 4, st, p);
        else KALLE_LAUNCH(attn_decode_kernel<0>, grid, block, (size_t)Nk * 4, st, p);
        return kalle_check_launch();
    }
    constexpr int lds = 3 * AT_TILE + 160 * 4 + 2 * 32 * AT_STRIDE;     // Q | K | V tiles, key bias [128 + 32], folded tail tiles
    static std::atomic<uint64_t> lds_ok{0};
    kalle_allow_lds(reinterpret_cast<const void*>(attn_fwd_kernel<1>), lds, lds_ok);
    dim3 grid((Nq + 127) / 128, H, B), block(512);
    KALLE_LAUNCH(attn_fwd_kernel<1>, grid, block, lds, static_cast<hipStream_t>(stream), p);
    return kalle_check_launch();
}

extern "C" int kalle_attention_bwd(const void* q, int64_t ldq, int q_off, const void* k, int64_t ldk, int k_off,
                                   const void* v, int64_t ldv, int v_off, const void* out, const void* dout,
                                   int64_t ldo, const float* lse, float* delta, void* dq, void* dk, void* dv,
                                   const float* rope_cos, const float* rope_sin, int rot, const uint8_t* key_mask,
                                   int causal, int B, int H, int Hkv, int Nq, int Nk, void* stream) {
    if (!out || !dout || !lse || !delta || !dq || !dk || !dv ||
        !check_common(q, ldq, q_off, k, ldk, k_off, v, ldv, v_off, ldo, rot, B, H, Hkv, Nq, Nk))
        return KALLE_ERR_ARG;
    if (rot && (!rope_cos || !rope_sin)) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    AttnParams p{};
    p.q = static_cast<const bf16_t*>(q); p.ldq = ldq; p.q_off = q_off;
    p.k = static_cast<const bf16_t*>(k); p.ldk = ldk; p.k_off = k_off;
    p.v = static_cast<const bf16_t*>(v); p.ldv = ldv; p.v_off = v_off;
    p.out = static_cast<bf16_t*>(const_cast<void*>(out)); p.ldo = ldo; p.lse = const_cast<float*>(lse);   // (read only here)
    p.cosT = rope_cos; p.sinT = rope_sin; p.rot = rot; p.mask = key_mask;
    p.B = B; p.H = H; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.causal = causal;
    if (causal && Nk < Nq) return KALLE_ERR_ARG;
    p.qpos = causal ? Nk - Nq : 0;
    p.dout = static_cast<const bf16_t*>(dout); p.delta = delta;
    p.dq = static_cast<bf16_t*>(dq); p.dk = static_cast<bf16_t*>(dk); p.dv = static_cast<bf16_t*>(dv);
    p.stamps = g_attn_stamps;

    // one block of queries and keys, one kv head per query head (the DiT's self-attention): everything in one kernel
    static const bool fused_env = !(getenv("KALLE_ATTN_FUSED_BWD") && atoi(getenv("KALLE_ATTN_FUSED_BWD")) == 0);
    if (fused_env && !causal && Nq <= 128 && Nk <= 128 && H == Hkv) {
        constexpr int flds = 4 * AT_TILE;
        static std::atomic<uint64_t> lds_ok_f{0};
        kalle_allow_lds(reinterpret_cast<const void*>(attn_bwd_fused_kernel), flds, lds_ok_f);
        KALLE_LAUNCH(attn_bwd_fused_kernel, dim3(1, H, B), dim3(512), flds, st, p);
        return kalle_check_launch();
    }
    // cross-attention of the DiT: several query heads per kv head and / or a few keys beyond one block, no rotary
    static const bool gqa_env = !(getenv("KALLE_ATTN_FUSED_GQA") && atoi(getenv("KALLE_ATTN_FUSED_GQA")) == 0);
    const int tail = Nk > 128 ? Nk - 128 : 0;
    if (fused_env && gqa_env && !causal && rot == 0 && Nq <= 128 && tail <= 16 && tail <= 128 - Nq) {
        constexpr int flds = 4 * AT_TILE;
        static std::atomic<uint64_t> lds_ok_g{0};
        kalle_allow_lds(reinterpret_cast<const void*>(attn_bwd_fused_gqa_kernel), flds, lds_ok_g);
        KALLE_LAUNCH(attn_bwd_fused_gqa_kernel, dim3(1, Hkv, B), dim3(512), flds, st, p);
        return kalle_check_launch();
    }
    constexpr int lds = 2 * AT_TILE + 256 * 4;
    static std::atomic<uint64_t> lds_ok_kv{0}, lds_ok_q{0};
    kalle_allow_lds(reinterpret_cast<const void*>(attn_bwd_kernel<true, 1>), lds, lds_ok_kv);
    kalle_allow_lds(reinterpret_cast<const void*>(attn_bwd_kernel<false, 1>), lds, lds_ok_q);
    // dQ first: it also produces delta, which the dK/dV kernel streams
    KALLE_LAUNCH((attn_bwd_kernel<false, 1>), dim3((Nq + 127) / 128, H, B), dim3(512), lds, st, p);
    KALLE_LAUNCH((attn_bwd_kernel<true, 1>), dim3((Nk + 127) / 128, Hkv, B), dim3(512), lds, st, p);
    return kalle_check_launch();
}
