// Off-default options of the DiT's transformer (gfx950): the depthwise convolution of the ConformerModule and the broadcast
// add of a position-embedding table.  Reference: stable_audio_tools/models/transformer.py:550-583 (ConformerModule: ... ->
// depthwise Conv1d(dim, dim, 17, groups = dim, padding = 8, bias = False) along the sequence -> ...), 45-87 + 796-797
// (x = x + pos_emb(x), one [n][dim] table for every batch element).
//
// Layout: the block keeps tokens row-major ([B][N][D], D contiguous - the GEMMs' layout), so the "channels" of the depthwise
// conv are the contiguous axis: a lane owns two neighbouring channels (one 4-byte bf16 pair per position, a wave reads 256
// contiguous bytes of a row), a workgroup of 256 lanes 512 channels x DW_TN positions; the K taps of its channels sit in LDS,
// transposed to [tap][channel] (conflict-free 8-byte reads).  The rows of the window are re-read through L1 tap by tap
// (DW_TN + K - 1 rows of 1 KiB per workgroup): HBM sees every input once.  None of this is on the DiT path the configs
// select - correctness first, no tuning.
#include <algorithm>

#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

constexpr int DW_TN = 16;        // output positions per workgroup
constexpr int DW_KMAX = 32;      // taps (the ConformerModule has 17)

// x[b][i] += t[i]
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ x, const float* __restrict__ t, int64_t n4,
                                                       int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        const f32x4 a = reinterpret_cast<const f32x4*>(t)[i % n4];
        v += a;
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

// y[b][n][c] = sum_k w[c][FLIP ? K-1-k : k] * x[b][n + k - pad][c]     (zero outside 0 <= n + k - pad < N)
// FLIP with pad' = K - 1 - pad is the data gradient of the un-flipped convolution.
template <bool OUT_BF16, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, void* __restrict__ y,
                                                     int N, int D, int K, int pad) {
    __shared__ float ws[DW_KMAX][512];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * 512;
    const int b = blockIdx.z, n0 = blockIdx.x * DW_TN;
    for (int i = tid; i < 512 * K; i += 256) {
        const int c = i / K, k = i - c * K;
        ws[FLIP ? K - 1 - k : k][c] = c0 + c < D ? w[(int64_t)(c0 + c) * K + k] : 0.f;
    }
    __syncthreads();
    const int c = c0 + 2 * tid;
    if (c >= D) return;                                  // (D is even: both channels of the pair exist)
    const bf16_t* xb = x + (int64_t)b * N * D + c;
    const int nend = min(n0 + DW_TN, N);
    for (int n = n0; n < nend; ++n) {
        float a0 = 0.f, a1 = 0.f;
        const int klo = max(0, pad - n), khi = min(K, N + pad - n);
        for (int k = klo; k < khi; ++k) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(xb + (int64_t)(n + k - pad) * D);
            const f32x2 wk = *reinterpret_cast<const f32x2*>(&ws[k][2 * tid]);
            a0 = fmaf(wk[0], bf16lo(v), a0);
            a1 = fmaf(wk[1], bf16hi(v), a1);
        }
        const int64_t o = ((int64_t)b * N + n) * D + c;
        if constexpr (OUT_BF16) *reinterpret_cast<uint32_t*>(static_cast<bf16_t*>(y) + o) = pack_bf16x2(a0, a1);
        else *reinterpret_cast<f32x2*>(static_cast<float*>(y) + o) = f32x2{a0, a1};
    }
}

// dw[c][k] += sum_{n in this workgroup's rows} dy[b][n][c] * x[b][n + k - pad][c]; one workgroup per (batch element, 512
// channels, DW_WN rows), one atomic add per (channel, tap) and workgroup
constexpr int DW_WN = 128;
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           float* __restrict__ dw, int N, int D, int K, int pad) {
    const int tid = threadIdx.x;
    const int c = blockIdx.y * 512 + 2 * tid;
    if (c >= D) return;
    const int b = blockIdx.z, n0 = blockIdx.x * DW_WN, nend = min(n0 + DW_WN, N);
    const bf16_t* xb = x + (int64_t)b * N * D + c;
    const bf16_t* gb = dy + (int64_t)b * N * D + c;
    for (int k = 0; k < K; ++k) {
        float s0 = 0.f, s1 = 0.f;
        const int lo = max(n0, pad - k), hi = min(nend, N + pad - k);
        for (int n = lo; n < hi; ++n) {
            const uint32_t g = *reinterpret_cast<const uint32_t*>(gb + (int64_t)n * D);
            const uint32_t v = *reinterpret_cast<const uint32_t*>(xb + (int64_t)(n + k - pad) * D);
            s0 = fmaf(bf16lo(g), bf16lo(v), s0);
            s1 = fmaf(bf16hi(g), bf16hi(v), s1);
        }
        atomicAdd(dw + (int64_t)c * K + k, s0);
        atomicAdd(dw + (int64_t)(c + 1) * K + k, s1);
    }
}

}  // namespace

extern "C" int kalle_add_rows(float* x, const float* table, int64_t nbatch, int64_t n, void* stream) {
    if (!x || !table || nbatch <= 0 || n <= 0 || (n & 3)) return KALLE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(table)) & 15) return KALLE_ERR_ARG;
    const int64_t total4 = nbatch * n / 4;
    const int grid = (int)std::min<int64_t>((total4 + 255) / 256, 2048);
    KALLE_LAUNCH(add_rows_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x, table, n / 4, total4);
    return kalle_check_launch();
}

static int dw_args_ok(const void* a, const void* b, const void* c, int B, int N, int D, int K, int pad) {
    if (!a || !b || !c || B <= 0 || N <= 0 || D <= 0 || (D & 1) || K <= 0 || K > DW_KMAX || pad < 0 || pad >= K) return 0;
    if (B > 65535 || (D + 511) / 512 > 65535) return 0;
    return 1;
}

extern "C" int kalle_dwconv1d_fwd(const void* x, const float* w, void* y, int y_dtype, int B, int N, int D, int K, int pad,
                                  int flip, void* stream) {
    if (!dw_args_ok(x, w, y, B, N, D, K, pad) || (y_dtype != KALLE_F32 && y_dtype != KALLE_BF16)) return KALLE_ERR_ARG;
    const dim3 grid((N + DW_TN - 1) / DW_TN, (D + 511) / 512, B), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bf16_t* xp = static_cast<const bf16_t*>(x);
    if (y_dtype == KALLE_BF16) {
        if (flip) KALLE_LAUNCH((dwconv_kernel<true, true>), grid, block, 0, st, xp, w, y, N, D, K, pad);
        else KALLE_LAUNCH((dwconv_kernel<true, false>), grid, block, 0, st, xp, w, y, N, D, K, pad);
    } else {
        if (flip) KALLE_LAUNCH((dwconv_kernel<false, true>), grid, block, 0, st, xp, w, y, N, D, K, pad);
        else KALLE_LAUNCH((dwconv_kernel<false, false>), grid, block, 0, st, xp, w, y, N, D, K, pad);
    }
    return kalle_check_launch();
}

extern "C" int kalle_dwconv1d_wgrad(const void* dy, const void* x, float* dw, int B, int N, int D, int K, int pad,
                                    void* stream) {
    if (!dw_args_ok(dy, x, dw, B, N, D, K, pad)) return KALLE_ERR_ARG;
    const dim3 grid((N + DW_WN - 1) / DW_WN, (D + 511) / 512, B), block(256);
    KALLE_LAUNCH(dwconv_wgrad_kernel, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(dy),
                 static_cast<const bf16_t*>(x), dw, N, D, K, pad);
    return kalle_check_launch();
}
