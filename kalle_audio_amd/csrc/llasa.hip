// Kernels of the Llasa task model's head and tail (model_sigmaVAE.py:53-104) around the Llama decoder layers (which run on
// the shared GEMM / attention / RMSNorm kernels): fixed-sigma latent sampling, token-embedding gather mixed with the
// projected audio latents under the two row masks (+ its scatter-add backward), exact GELU, and the masked fixed-sigma
// Gaussian KL losses.  All HBM-bound: vectorised where rows are long, fp32 math.
#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

inline int grid_for(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// out = a * x + b * y   (model_sigmaVAE.py:166: x = mean + std * randn_like(mean))
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                    float* __restrict__ out, float a, float b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = a * x[i] + b * y[i];
}

// single-row GEMM for KV-cached decoding (model_sigmaVAE.py:122-146 with a cache): y[n] = sum_k W[n][k] x[k] (+ residual[n]).
// Pure weight streaming: a wave owns 2 weight rows, its lanes walk K in 16-byte chunks (x staged once per workgroup in
// LDS), shuffle-reduce, lane 0 writes.  8 rows per 256-thread workgroup -> N/8 workgroups keep every HBM channel busy.
template <bool YF32>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W, int64_t ldw,
                                                   void* __restrict__ y, const float* __restrict__ res, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(gsm);
    for (int i = threadIdx.x; i < (K >> 3); i += 256)
        reinterpret_cast<i32x4*>(xs)[i] = reinterpret_cast<const i32x4*>(x)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 8 + wave * 2;
    if (n0 >= N) return;
    const bool two = n0 + 1 < N;
    const bf16_t* w0 = W + (int64_t)n0 * ldw;
    const bf16_t* w1 = W + (int64_t)(two ? n0 + 1 : n0) * ldw;
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane; c < (K >> 3); c += 64) {
        const i32x4 xv = reinterpret_cast<const i32x4*>(xs)[c];
        const i32x4 u = *reinterpret_cast<const i32x4*>(w0 + 8 * c);
        const i32x4 v = *reinterpret_cast<const i32x4*>(w1 + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = bf16lo((uint32_t)xv[e]), x1 = bf16hi((uint32_t)xv[e]);
            a0 += bf16lo((uint32_t)u[e]) * x0 + bf16hi((uint32_t)u[e]) * x1;
            a1 += bf16lo((uint32_t)v[e]) * x0 + bf16hi((uint32_t)v[e]) * x1;
        }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
        if (res) { a0 += res[n0]; if (two) a1 += res[n0 + 1]; }
        if constexpr (YF32) {
            static_cast<float*>(y)[n0] = a0;
            if (two) static_cast<float*>(y)[n0 + 1] = a1;
        } else {
            static_cast<bf16_t*>(y)[n0] = f32_to_bf16(a0);
            if (two) static_cast<bf16_t*>(y)[n0 + 1] = f32_to_bf16(a1);
        }
    }
}

// peak normalisation to int16 (infer_0723.py:293: x / max|x| -> clamp(-1, 1) * 32767 -> int16, truncating like .to(int16))
template <bool F32>
__global__ __launch_bounds__(256) void absmax_kernel(const void* __restrict__ x, unsigned* __restrict__ peak_bits, int64_t n) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(F32 ? static_cast<const float*>(x)[i] : bf16_to_f32(static_cast<const bf16_t*>(x)[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)   // non-negative floats order like their bit patterns
        atomicMax(peak_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}
template <bool F32>
__global__ __launch_bounds__(256) void to_int16_kernel(const void* __restrict__ x, const unsigned* __restrict__ peak_bits,
                                                       int16_t* __restrict__ out, int64_t n) {
    const float peak = __uint_as_float(peak_bits[0]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = __fdiv_rn(F32 ? static_cast<const float*>(x)[i] : bf16_to_f32(static_cast<const bf16_t*>(x)[i]), peak);
        v = fminf(fmaxf(v, -1.f), 1.f) * 32767.f;
        out[i] = (int16_t)(int)v;
    }
}

// out[r, :] = audio[r, :] * am[r] + table[ids[r], :] * im[r]     (model_sigmaVAE.py:66, 73)
template <bool AF32>
__global__ __launch_bounds__(256) void embed_mix_fwd_kernel(const int64_t* __restrict__ ids,
                                                            const float* __restrict__ table,
                                                            const void* __restrict__ audio,
                                                            const float* __restrict__ im, const float* __restrict__ am,
                                                            float* __restrict__ out, int64_t rows, int D, int64_t V) {
    const int cpr = D >> 2;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * 4;
        const float wi = im[r], wa = am[r];
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (wi != 0.f) {
            int64_t id = ids[r];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            const f32x4 e = *reinterpret_cast<const f32x4*>(table + id * D + c);
            o = e * wi;
        }
        if (wa != 0.f) {
            f32x4 a;
            if constexpr (AF32) {
                a = *reinterpret_cast<const f32x4*>(static_cast<const float*>(audio) + r * D + c);
            } else {
                const i32x2 v = *reinterpret_cast<const i32x2*>(static_cast<const bf16_t*>(audio) + r * D + c);
                a = f32x4{bf16lo((uint32_t)v[0]), bf16hi((uint32_t)v[0]), bf16lo((uint32_t)v[1]), bf16hi((uint32_t)v[1])};
            }
            o += a * wa;
        }
        *reinterpret_cast<f32x4*>(out + r * D + c) = o;
    }
}

// daudio[r, :] = dout[r, :] * am[r] ;  dtable[ids[r], :] += dout[r, :] * im[r]  (fp32 atomics: tokens repeat)
__global__ __launch_bounds__(256) void embed_mix_bwd_kernel(const float* __restrict__ dout,
                                                            const int64_t* __restrict__ ids,
                                                            const float* __restrict__ im, const float* __restrict__ am,
                                                            float* __restrict__ dtable, float* __restrict__ daudio,
                                                            int64_t rows, int D, int64_t V) {
    const int cpr = D >> 2;
    const int64_t total = rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr) * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(dout + r * D + c);
        if (daudio) *reinterpret_cast<f32x4*>(daudio + r * D + c) = g * am[r];
        const float wi = im[r];
        if (dtable && wi != 0.f) {
            int64_t id = ids[r];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            float* dst = dtable + id * D + c;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dst + e, g[e] * wi);
        }
    }
}

// exact GELU (nn.GELU() default, model_sigmaVAE.py:46): 0.5 x (1 + erf(x / sqrt 2))
template <bool F32>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = F32 ? static_cast<const float*>(x)[i] : bf16_to_f32(static_cast<const bf16_t*>(x)[i]);
        const float o = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
        if constexpr (F32) static_cast<float*>(y)[i] = o;
        else static_cast<bf16_t*>(y)[i] = f32_to_bf16(o);
    }
}
template <bool F32>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                       void* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = F32 ? static_cast<const float*>(x)[i] : bf16_to_f32(static_cast<const bf16_t*>(x)[i]);
        const float g = F32 ? static_cast<const float*>(dy)[i] : bf16_to_f32(static_cast<const bf16_t*>(dy)[i]);
        const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752f));
        const float pdf = 0.39894228040143268f * __expf(-0.5f * v * v);
        const float o = g * (cdf + v * pdf);
        if constexpr (F32) static_cast<float*>(dx)[i] = o;
        else static_cast<bf16_t*>(dx)[i] = f32_to_bf16(o);
    }
}

// KL( N(pred, s) || N(label, s) ) = (pred - label)^2 / (2 s^2), summed over the latent dim / dim, then the two masked
// sums over rows (model_sigmaVAE.py:85-95).  One wave per row; sums[0..3] += {kl*ma, ma, kl*mb, mb}.
__global__ __launch_bounds__(256) void gauss_kl_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ label,
                                                           const float* __restrict__ ma, const float* __restrict__ mb,
                                                           float* __restrict__ sums, float coef, int64_t rows, int d) {
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < d; c += 64) {
            const float t = pred[r * d + c] - label[r * d + c];
            s += t * t;
        }
        s = wave_sum(s) * coef;
        const float a = ma[r], b = mb[r];
        acc[0] += s * a; acc[1] += a; acc[2] += s * b; acc[3] += b;
    }
    if (lane == 0)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(sums + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                                       red[3][threadIdx.x]);
}
// dpred[r, c] = 2 coef (pred - label) * (ga * ma[r] / sum(ma) + gb * mb[r] / sum(mb)); ga / gb: upstream gradients of the
// two losses (device scalars), sums from the forward
__global__ __launch_bounds__(256) void gauss_kl_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ label,
                                                           const float* __restrict__ ma, const float* __restrict__ mb,
                                                           const float* __restrict__ sums, const float* __restrict__ ga,
                                                           const float* __restrict__ gb, float* __restrict__ dpred,
                                                           float coef, int64_t rows, int d) {
    const float wa = ga[0] / sums[1], wb = gb[0] / sums[3];
    const int64_t total = rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        dpred[i] = 2.f * coef * (pred[i] - label[i]) * (wa * ma[r] + wb * mb[r]);
    }
}

}  // namespace

extern "C" int kalle_axpby(const float* x, const float* y, float* out, float a, float b, int64_t n, void* stream) {
    if (!x || !y || !out || n <= 0) return KALLE_ERR_ARG;
    KALLE_LAUNCH(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, out, a, b, n);
    return kalle_check_launch();
}

extern "C" int kalle_gemv_bf16(const void* x, const void* W, int64_t ldw, void* y, int y_dtype, const float* residual,
                               int N, int K, void* stream) {
    if (!x || !W || !y || N <= 0 || K <= 0 || (K & 7) || (ldw & 7) || K > 32768) return KALLE_ERR_ARG;
    if (y_dtype != KALLE_F32 && y_dtype != KALLE_BF16) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((N + 7) / 8);
    if (y_dtype == KALLE_F32)
        KALLE_LAUNCH((gemv_kernel<true>), grid, dim3(256), (size_t)K * 2, st, static_cast<const bf16_t*>(x),
                     static_cast<const bf16_t*>(W), ldw, y, residual, N, K);
    else
        KALLE_LAUNCH((gemv_kernel<false>), grid, dim3(256), (size_t)K * 2, st, static_cast<const bf16_t*>(x),
                     static_cast<const bf16_t*>(W), ldw, y, residual, N, K);
    return kalle_check_launch();
}

extern "C" int kalle_peak_normalize_int16(const void* x, int dtype, float* peak, int16_t* out, int64_t n, void* stream) {
    if (!x || !peak || !out || n <= 0) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(peak, 0, sizeof(float), st) != hipSuccess) return KALLE_ERR_LAUNCH;
    unsigned* pb = reinterpret_cast<unsigned*>(peak);
    if (dtype == KALLE_F32) {
        KALLE_LAUNCH((absmax_kernel<true>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, pb, n);
        KALLE_LAUNCH((to_int16_kernel<true>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, pb, out, n);
    } else {
        KALLE_LAUNCH((absmax_kernel<false>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, pb, n);
        KALLE_LAUNCH((to_int16_kernel<false>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, pb, out, n);
    }
    return kalle_check_launch();
}

extern "C" int kalle_embed_mix_fwd(const int64_t* ids, const float* table, const void* audio, int audio_dtype,
                                   const float* ids_mask, const float* audio_mask, float* out, int64_t rows, int D,
                                   int64_t vocab, void* stream) {
    if (!ids || !table || !audio || !ids_mask || !audio_mask || !out || rows <= 0 || D <= 0 || (D & 3) || vocab <= 0)
        return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(rows * (D >> 2), 256));
    if (audio_dtype == KALLE_F32)
        KALLE_LAUNCH((embed_mix_fwd_kernel<true>), grid, dim3(256), 0, st, ids, table, audio, ids_mask, audio_mask, out, rows,
                     D, vocab);
    else
        KALLE_LAUNCH((embed_mix_fwd_kernel<false>), grid, dim3(256), 0, st, ids, table, audio, ids_mask, audio_mask, out,
                     rows, D, vocab);
    return kalle_check_launch();
}

extern "C" int kalle_embed_mix_bwd(const float* dout, const int64_t* ids, const float* ids_mask, const float* audio_mask,
                                   float* dtable, float* daudio, int64_t rows, int D, int64_t vocab, void* stream) {
    if (!dout || !ids || !ids_mask || !audio_mask || (!dtable && !daudio) || rows <= 0 || D <= 0 || (D & 3) || vocab <= 0)
        return KALLE_ERR_ARG;
    KALLE_LAUNCH(embed_mix_bwd_kernel, dim3(grid_for(rows * (D >> 2), 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                 dout, ids, ids_mask, audio_mask, dtable, daudio, rows, D, vocab);
    return kalle_check_launch();
}

extern "C" int kalle_gelu_fwd(const void* x, void* y, int dtype, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KALLE_F32) KALLE_LAUNCH((gelu_fwd_kernel<true>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, y, n);
    else KALLE_LAUNCH((gelu_fwd_kernel<false>), dim3(grid_for(n, 256)), dim3(256), 0, st, x, y, n);
    return kalle_check_launch();
}

extern "C" int kalle_gelu_bwd(const void* dy, const void* x, void* dx, int dtype, int64_t n, void* stream) {
    if (!dy || !x || !dx || n <= 0) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KALLE_F32) KALLE_LAUNCH((gelu_bwd_kernel<true>), dim3(grid_for(n, 256)), dim3(256), 0, st, dy, x, dx, n);
    else KALLE_LAUNCH((gelu_bwd_kernel<false>), dim3(grid_for(n, 256)), dim3(256), 0, st, dy, x, dx, n);
    return kalle_check_launch();
}

extern "C" int kalle_gauss_kl_fwd(const float* pred, const float* label, const float* mask_a, const float* mask_b,
                                  float* sums4, float std, int64_t rows, int dim, void* stream) {
    if (!pred || !label || !mask_a || !mask_b || !sums4 || rows <= 0 || dim <= 0 || !(std > 0.f)) return KALLE_ERR_ARG;
    const float coef = 1.f / (2.f * std * std * (float)dim);
    KALLE_LAUNCH(gauss_kl_fwd_kernel, dim3(grid_for((rows + 3) / 4, 1)), dim3(256), 0, static_cast<hipStream_t>(stream), pred,
                 label, mask_a, mask_b, sums4, coef, rows, dim);
    return kalle_check_launch();
}

extern "C" int kalle_gauss_kl_bwd(const float* pred, const float* label, const float* mask_a, const float* mask_b,
                                  const float* sums4, const float* grad_a, const float* grad_b, float* dpred, float std,
                                  int64_t rows, int dim, void* stream) {
    if (!pred || !label || !mask_a || !mask_b || !sums4 || !grad_a || !grad_b || !dpred || rows <= 0 || dim <= 0 ||
        !(std > 0.f))
        return KALLE_ERR_ARG;
    const float coef = 1.f / (2.f * std * std * (float)dim);
    KALLE_LAUNCH(gauss_kl_bwd_kernel, dim3(grid_for(rows * dim, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), pred,
                 label, mask_a, mask_b, sums4, grad_a, grad_b, dpred, coef, rows, dim);
    return kalle_check_launch();
}
