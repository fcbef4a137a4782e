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
// The LDS copy of x can be built on the fly so that the decode step needs no separate norm / activation launches:
//   PRO_RMS:    x fp32 [K] -> bf16(x * (gamma * rsqrt(mean(x^2) + eps)))   (LlamaRMSNorm, same rounding as rms_fwd_kernel)
//   PRO_SWIGLU: x bf16 [2K] = [up | gate] -> bf16(up * silu(gate))         (LlamaMLP, same rounding as swiglu_fwd_kernel)
// Rows >= nsplit go to y2 (the k | v part of the fused qkv projection lands directly in its KV-cache row).
enum { PRO_BF16 = 0, PRO_RMS = 1, PRO_SWIGLU = 2 };

template <bool YF32, int PRO>
__global__ __launch_bounds__(256) void gemv_kernel(const void* __restrict__ xin, const float* __restrict__ gamma, float eps,
                                                   const bf16_t* __restrict__ W, int64_t ldw, void* __restrict__ y,
                                                   void* __restrict__ y2, int nsplit, const float* __restrict__ res,
                                                   int N, int K, int rpw) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(gsm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the weight stream does not depend on x: the first PF chunks of both rows are in flight while x is being prepared.
    // A wave walks `rpw` row pairs (rows ((blockIdx * rpw + p) * 4 + wave) * 2 + {0, 1}); its work is the flat sequence
    // of (pair, K batch) items, each item's loads issued one item ahead of its FMAs.
    constexpr int PF = 4;
    const int nc = K >> 3;
    const int NB = (nc + 64 * PF - 1) / (64 * PF);
    const int T = rpw * NB;
    auto load = [&](int pr, int bb, i32x4* uu, i32x4* vv) {
        const int n = ((blockIdx.x * rpw + pr) * 4 + wave) * 2;
        const bf16_t* r0 = W + (int64_t)(n < N ? n : N - 1) * ldw;
        const bf16_t* r1 = W + (int64_t)(n + 1 < N ? n + 1 : N - 1) * ldw;
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int c = bb * (64 * PF) + lane + 64 * j;
            const int cc = c < nc ? c : nc - 1;
            uu[j] = *reinterpret_cast<const i32x4*>(r0 + 8 * cc);
            vv[j] = *reinterpret_cast<const i32x4*>(r1 + 8 * cc);
        }
    };
    i32x4 u[PF], v[PF];
    load(0, 0, u, v);
    if constexpr (PRO == PRO_BF16) {
        const bf16_t* x = static_cast<const bf16_t*>(xin);
        for (int i = threadIdx.x; i < (K >> 3); i += 256)
            reinterpret_cast<i32x4*>(xs)[i] = reinterpret_cast<const i32x4*>(x)[i];
    } else if constexpr (PRO == PRO_RMS) {
        __shared__ float red[4];
        const float* x = static_cast<const float*>(xin);
        float q = 0.f;
        for (int i = threadIdx.x; i < (K >> 2); i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
            q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        q = wave_sum(q);
        if (lane == 0) red[wave] = q;
        __syncthreads();
        const float rr = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)K + eps);
        for (int i = threadIdx.x; i < (K >> 2); i += 256) {
            const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
            const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[i];
            i32x2 o;
            o[0] = (int)pack_bf16x2(v[0] * (g[0] * rr), v[1] * (g[1] * rr));
            o[1] = (int)pack_bf16x2(v[2] * (g[2] * rr), v[3] * (g[3] * rr));
            reinterpret_cast<i32x2*>(xs)[i] = o;
        }
    } else {
        const bf16_t* h = static_cast<const bf16_t*>(xin);
        for (int i = threadIdx.x; i < (K >> 3); i += 256) {
            const i32x4 xv = reinterpret_cast<const i32x4*>(h)[i];
            const i32x4 gv = reinterpret_cast<const i32x4*>(h + K)[i];
            i32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] = (int)pack_bf16x2(bf16lo((uint32_t)xv[j]) * siluf_(bf16lo((uint32_t)gv[j])),
                                        bf16hi((uint32_t)xv[j]) * siluf_(bf16hi((uint32_t)gv[j])));
            reinterpret_cast<i32x4*>(xs)[i] = o;
        }
    }
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    int pr = 0, bb = 0;
    for (int t = 0; t < T; ++t) {
        i32x4 un[PF], vn[PF];
        const bool more = t + 1 < T;
        const int npr = bb + 1 == NB ? pr + 1 : pr, nbb = bb + 1 == NB ? 0 : bb + 1;
        if (more) load(npr, nbb, un, vn);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int c = bb * (64 * PF) + lane + 64 * j;
            if (c < nc) {
                const i32x4 xv = reinterpret_cast<const i32x4*>(xs)[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x0 = bf16lo((uint32_t)xv[e]), x1 = bf16hi((uint32_t)xv[e]);
                    a0 += bf16lo((uint32_t)u[j][e]) * x0 + bf16hi((uint32_t)u[j][e]) * x1;
                    a1 += bf16lo((uint32_t)v[j][e]) * x0 + bf16hi((uint32_t)v[j][e]) * x1;
                }
            }
        }
        if (bb + 1 == NB) {   // row pair done
            const int n0 = ((blockIdx.x * rpw + pr) * 4 + wave) * 2;
            a0 = wave_sum(a0);
            a1 = wave_sum(a1);
            if (lane == 0 && n0 < N) {
                const bool two = n0 + 1 < N;
                if (res) { a0 += res[n0]; if (two) a1 += res[n0 + 1]; }
                // nsplit is even (a multiple of 64 in practice), so a wave's two rows never straddle it
                void* yo = n0 < nsplit ? y : y2;
                const int r0 = n0 < nsplit ? n0 : n0 - nsplit;
                if constexpr (YF32) {
                    static_cast<float*>(yo)[r0] = a0;
                    if (two) static_cast<float*>(yo)[r0 + 1] = a1;
                } else {
                    static_cast<bf16_t*>(yo)[r0] = f32_to_bf16(a0);
                    if (two) static_cast<bf16_t*>(yo)[r0 + 1] = f32_to_bf16(a1);
                }
            }
            a0 = a1 = 0.f;
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < PF; ++j) { u[j] = un[j]; v[j] = vn[j]; }
        }
        pr = npr; bb = nbb;
    }
}

template <bool YF32, int PRO>
inline void gemv_launch(const void* x, const float* gamma, float eps, const void* W, int64_t ldw, void* y, void* y2,
                        int nsplit, const float* res, int N, int K, hipStream_t st) {
    // several row pairs per wave once there are enough workgroups: the per-workgroup x preparation is amortised
    const int rpw = N >= 8 * 4 * 512 ? 4 : N >= 8 * 2 * 512 ? 2 : 1;
    KALLE_LAUNCH((gemv_kernel<YF32, PRO>), dim3((N + 8 * rpw - 1) / (8 * rpw)), dim3(256), (size_t)K * 2, st, x, gamma,
                 eps, static_cast<const bf16_t*>(W), ldw, y, y2, nsplit, res, N, K, rpw);
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

// ---- two-Gaussian KL of model.py:84-100 --------------------------------------------------------------------------------
// KL( N(m1, s1) || N(m2, s2) ) = log(s2 / s1) + (s1^2 + (m1 - m2)^2) / (2 s2^2) - 1/2 per latent element, with
// m2 | log s2 = the two halves of the head's output row [2 dim]; summed over the latent dim / dim; two masked sums over rows.
// Label statistics: MODE 0 - explicit mean / std rows [dim] (a caller-supplied transform already applied);
//                   MODE 1 - one row [2 dim] = mean | scale of the Oobleck encoder, std = softplus(scale) + 1e-4
//                            (stable_audio_tools/models/bottleneck.py:51-54 on the same tensor); both times std_mult (1.25).
template <int MODE>
__device__ __forceinline__ void kl2_label(const float* __restrict__ lmean, const float* __restrict__ lstd, int64_t r, int c,
                                          int d, float std_mult, float& m1, float& s1) {
    if constexpr (MODE == 0) {
        m1 = lmean[r * d + c];
        s1 = lstd[r * d + c] * std_mult;
    } else {
        m1 = lmean[r * 2 * d + c];
        const float sc = lmean[r * 2 * d + d + c];
        s1 = ((sc > 20.f ? sc : log1pf(expf(sc))) + 1e-4f) * std_mult;      // F.softplus (threshold 20)
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void gauss_kl2_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ lmean,
                                                            const float* __restrict__ lstd, const float* __restrict__ ma,
                                                            const float* __restrict__ mb, float* __restrict__ sums,
                                                            float std_mult, int64_t rows, int d) {
    __shared__ float red[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float inv_d = 1.f / (float)d;
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane; c < d; c += 64) {
            float m1, s1;
            kl2_label<MODE>(lmean, lstd, r, c, d, std_mult, m1, s1);
            const float m2 = pred[r * 2 * d + c], l2 = pred[r * 2 * d + d + c];
            const float dm = m1 - m2;
            s += l2 - logf(s1) + 0.5f * (s1 * s1 + dm * dm) * expf(-2.f * l2) - 0.5f;
        }
        s = wave_sum(s) * inv_d;
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

// d kl / d m2 = (m2 - m1) / s2^2 ;  d kl / d log s2 = 1 - (s1^2 + (m1 - m2)^2) / s2^2 ; times the row weight / dim
template <int MODE>
__global__ __launch_bounds__(256) void gauss_kl2_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ lmean,
                                                            const float* __restrict__ lstd, const float* __restrict__ ma,
                                                            const float* __restrict__ mb, const float* __restrict__ sums,
                                                            const float* __restrict__ ga, const float* __restrict__ gb,
                                                            float* __restrict__ dpred, float std_mult, int64_t rows, int d) {
    const float wa = ga[0] / sums[1], wb = gb[0] / sums[3];
    const float inv_d = 1.f / (float)d;
    const int64_t total = rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        float m1, s1;
        kl2_label<MODE>(lmean, lstd, r, c, d, std_mult, m1, s1);
        const float m2 = pred[r * 2 * d + c], l2 = pred[r * 2 * d + d + c];
        const float w = (wa * ma[r] + wb * mb[r]) * inv_d, e = expf(-2.f * l2), dm = m2 - m1;
        dpred[r * 2 * d + c] = w * dm * e;
        dpred[r * 2 * d + d + c] = w * (1.f - (s1 * s1 + dm * dm) * e);
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
    if (y_dtype == KALLE_F32) gemv_launch<true, PRO_BF16>(x, nullptr, 0.f, W, ldw, y, y, N, residual, N, K, st);
    else gemv_launch<false, PRO_BF16>(x, nullptr, 0.f, W, ldw, y, y, N, residual, N, K, st);
    return kalle_check_launch();
}

extern "C" int kalle_llama_decode_ws_bytes(int H, int Hkv, int inner) {
    if (H <= 0 || Hkv <= 0 || inner <= 0) return KALLE_ERR_ARG;
    const int64_t D = (int64_t)H * 64;
    // x2 | x3 fp32, lse fp32 (padded), q | ao | hf bf16
    return (int)(2 * D * 4 + ((H * 4 + 63) & ~63) + D * 2 + D * 2 + 2 * (int64_t)inner * 2);
}

extern "C" int kalle_llama_decode_step(const kalle_llama_layer* layers, int n_layers, const float* x, float* out, int H,
                                       int Hkv, int inner, float eps, int t0, int cache_rows, const float* rope_cos,
                                       const float* rope_sin, void* workspace, void* stream) {
    if (!layers || n_layers <= 0 || !x || !out || !workspace || !rope_cos || !rope_sin) return KALLE_ERR_ARG;
    if (H <= 0 || Hkv <= 0 || H % Hkv || inner <= 0 || (inner & 7) || t0 < 0 || t0 >= cache_rows) return KALLE_ERR_ARG;
    const int D = H * 64, kvw = 2 * Hkv * 64;
    if (D > 32768 || inner > 32768) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace);
    float* xa = reinterpret_cast<float*>(ws);               // x2: residual stream after the attention branch
    float* xb = xa + D;                                     // x3: layer output (input of the next layer)
    float* lse = xb + D;
    bf16_t* q = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(lse) + ((H * 4 + 63) & ~63));
    bf16_t* ao = q + D;
    bf16_t* hf = ao + D;
    const float* xin = x;
    for (int l = 0; l < n_layers; ++l) {
        const kalle_llama_layer& L = layers[l];
        if (!L.input_norm || !L.wqkv || !L.wo || !L.post_norm || !L.wug || !L.wdown || !L.kv_cache) return KALLE_ERR_ARG;
        bf16_t* kv_row = static_cast<bf16_t*>(L.kv_cache) + (int64_t)t0 * kvw;
        // q -> scratch, k | v -> cache row t0 (un-rotated: the attention kernel rotates by row index)
        gemv_launch<false, PRO_RMS>(xin, L.input_norm, eps, L.wqkv, D, q, kv_row, D, nullptr, D + kvw, D, st);
        int rc = kalle_attention_fwd(q, D, 0, L.kv_cache, kvw, 0, L.kv_cache, kvw, Hkv * 64, ao, D, lse, rope_cos, rope_sin,
                                     64, nullptr, 1, 1, H, Hkv, 1, t0 + 1, stream);
        if (rc != KALLE_OK) return rc;
        gemv_launch<true, PRO_BF16>(ao, nullptr, 0.f, L.wo, D, xa, xa, D, xin, D, D, st);
        gemv_launch<false, PRO_RMS>(xa, L.post_norm, eps, L.wug, D, hf, hf, 2 * inner, nullptr, 2 * inner, D, st);
        float* xo = l + 1 == n_layers ? out : xb;
        gemv_launch<true, PRO_SWIGLU>(hf, nullptr, 0.f, L.wdown, inner, xo, xo, D, xa, D, inner, st);
        xin = xo;
    }
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

extern "C" int kalle_gauss_kl2_fwd(const float* pred, const float* label_mean, const float* label_std, int label_mode,
                                   float std_mult, const float* mask_a, const float* mask_b, float* sums4, int64_t rows,
                                   int dim, void* stream) {
    if (!pred || !label_mean || !mask_a || !mask_b || !sums4 || rows <= 0 || dim <= 0 || !(std_mult > 0.f)) return KALLE_ERR_ARG;
    if (label_mode != 0 && label_mode != 1) return KALLE_ERR_ARG;
    if (label_mode == 0 && !label_std) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for((rows + 3) / 4, 1)), block(256);
    if (label_mode == 0) KALLE_LAUNCH(gauss_kl2_fwd_kernel<0>, grid, block, 0, st, pred, label_mean, label_std, mask_a, mask_b, sums4, std_mult, rows, dim);
    else KALLE_LAUNCH(gauss_kl2_fwd_kernel<1>, grid, block, 0, st, pred, label_mean, label_std, mask_a, mask_b, sums4, std_mult, rows, dim);
    return kalle_check_launch();
}

extern "C" int kalle_gauss_kl2_bwd(const float* pred, const float* label_mean, const float* label_std, int label_mode,
                                   float std_mult, const float* mask_a, const float* mask_b, const float* sums4,
                                   const float* grad_a, const float* grad_b, float* dpred, int64_t rows, int dim,
                                   void* stream) {
    if (!pred || !label_mean || !mask_a || !mask_b || !sums4 || !grad_a || !grad_b || !dpred || rows <= 0 || dim <= 0 ||
        !(std_mult > 0.f))
        return KALLE_ERR_ARG;
    if (label_mode != 0 && label_mode != 1) return KALLE_ERR_ARG;
    if (label_mode == 0 && !label_std) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(grid_for(rows * dim, 256)), block(256);
    if (label_mode == 0) KALLE_LAUNCH(gauss_kl2_bwd_kernel<0>, grid, block, 0, st, pred, label_mean, label_std, mask_a, mask_b, sums4, grad_a, grad_b, dpred, std_mult, rows, dim);
    else KALLE_LAUNCH(gauss_kl2_bwd_kernel<1>, grid, block, 0, st, pred, label_mean, label_std, mask_a, mask_b, sums4, grad_a, grad_b, dpred, std_mult, rows, dim);
    return kalle_check_launch();
}
