// 1-D convolution stack of the audio VAEs (gfx950, vector ALUs only - no MFMA, per the north-star spec):
// weight-norm fold, SnakeBeta/ELU fused into the input staging, dilated / strided conv and transposed conv with
// LDS line buffers, residual add + tanh fused into the store.
// Reference: stable_audio_tools/models/autoencoders.py:39-62 (ResidualUnit), 64-81 (EncoderBlock), 83-114
// (DecoderBlock), 116-191 (Oobleck encoder/decoder); blocks.py:301-339 (SnakeBeta); dac.nn.layers.WNConv1d ->
// torch.nn.utils.weight_norm (w = g * v / ||v||, norm over all dims but 0).
//
// Layout: activations (B, C, L) row-major (L contiguous, as torch), fp32 or bf16; weights are repacked once per
// forward by kalle_weight_norm_fold into [Cin][K][Cout] fp32 so that a workgroup's weight slab is contiguous in Cout.
// Tile: 64 output channels x 64 output positions per 256-thread workgroup, 4x4 outputs per thread; input channels
// are streamed 8 at a time through an LDS line buffer that holds the activated input span (with halo) once.
#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

constexpr int CO_T = 64, L_T = 64, CI_T = 8;
constexpr int MAX_SPAN = 576;   // (L_T-1)*stride + (K-1)*dil + 1 must fit
constexpr int MAX_K = 16;

__device__ __forceinline__ float act_apply(float x, int act, float a, float inv_b) {
    if (act == 1) {  // SnakeBeta: x + sin^2(x*alpha)/(beta+1e-9)
        const float s = sinf(x * a);
        return x + inv_b * s * s;
    }
    if (act == 2) return x > 0.f ? x : (__expf(x) - 1.f);  // ELU(alpha=1)
    if (act == 3) return x > 0.f ? x : x * a;              // LeakyReLU(negative_slope = a)
    return x;
}

template <bool F32>
__device__ __forceinline__ float ld1(const void* p, int64_t i) {
    if constexpr (F32) return static_cast<const float*>(p)[i];
    else return bf16_to_f32(static_cast<const bf16_t*>(p)[i]);
}
template <bool F32>
__device__ __forceinline__ void st1(void* p, int64_t i, float v) {
    if constexpr (F32) static_cast<float*>(p)[i] = v;
    else static_cast<bf16_t*>(p)[i] = f32_to_bf16(v);
}

struct ConvParams {
    const void* x; const float* w; const float* bias; const void* res; void* y;
    int B, Cin, Lin, Cout, Lout, K, stride, pad, dil, act, post;
    const float* aa; const float* ab; int logscale;
    float act_param;
    float out_scale;
    int xC;   // channels of the x tensor (2*Cin for the gated activation, else Cin)
};

template <bool XF32, bool YF32>
__global__ __launch_bounds__(256) void conv1d_kernel(ConvParams p) {
    __shared__ float Xs[CI_T][MAX_SPAN];
    __shared__ __attribute__((aligned(16))) float Ws[CI_T][MAX_K][CO_T];
    const int tid = threadIdx.x;
    const int tc = tid & 15, tl = tid >> 4;           // 16 channel groups x 16 position groups
    const int b = blockIdx.z, co0 = blockIdx.y * CO_T, l0 = blockIdx.x * L_T;
    const int span = (L_T - 1) * p.stride + (p.K - 1) * p.dil + 1;
    const int in0 = l0 * p.stride - p.pad;            // input position of span element 0

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += CI_T) {
        __syncthreads();
        // stage activated input span: consecutive threads -> consecutive positions (coalesced)
        for (int idx = tid; idx < CI_T * span; idx += 256) {
            const int c = idx / span, sp = idx - c * span;
            const int ci = ci0 + c, li = in0 + sp;
            float v = 0.f;
            if (ci < p.Cin && li >= 0 && li < p.Lin) {
                const int64_t xi = ((int64_t)b * p.xC + ci) * p.Lin + li;
                v = ld1<XF32>(p.x, xi);
                if (p.act == 1) {
                    float a = p.aa[ci], bb = p.ab[ci];
                    if (p.logscale) { a = __expf(a); bb = __expf(bb); }
                    v = act_apply(v, 1, a, 1.f / (bb + 1e-9f));
                } else if (p.act == 4) {  // WaveNet gate: tanh(x[ci]) * sigmoid(x[Cin + ci])
                    const float gt = ld1<XF32>(p.x, xi + (int64_t)p.Cin * p.Lin);
                    v = tanhf(v) / (1.f + __expf(-gt));
                } else if (p.act >= 2) {
                    v = act_apply(v, p.act, p.act_param, 0.f);
                }
            }
            Xs[c][sp] = v;
        }
        // stage weights [ci][k][co] (contiguous in co)
        for (int idx = tid; idx < CI_T * p.K * CO_T; idx += 256) {
            const int co = idx & (CO_T - 1);
            const int ck = idx >> 6;
            const int c = ck / p.K, k = ck - c * p.K;
            const int ci = ci0 + c;
            float v = 0.f;
            if (ci < p.Cin && co0 + co < p.Cout) v = p.w[((int64_t)ci * p.K + k) * p.Cout + co0 + co];
            Ws[c][k][co] = v;
        }
        __syncthreads();
#pragma unroll 2
        for (int c = 0; c < CI_T; ++c) {
            for (int k = 0; k < p.K; ++k) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[c][k][4 * tc]);
                const int xb = (4 * tl) * p.stride + k * p.dil;
                float xv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = Xs[c][xb + j * p.stride];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(wv[i], xv[j], acc[i][j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + 4 * tc + i;
        if (co >= p.Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * tl + j;
            if (l >= p.Lout) continue;
            const int64_t oi = ((int64_t)b * p.Cout + co) * p.Lout + l;
            float v = acc[i][j] + bv;
            if (p.res) v += ld1<XF32>(p.res, oi);
            v *= p.out_scale;
            if (p.post & 2) v += ld1<YF32>(p.y, oi);   // accumulate into y (sum over parallel AMP blocks)
            if (p.post & 1) v = tanhf(v);
            st1<YF32>(p.y, oi, v);
        }
    }
}

// transposed conv: y[b,co,lo] = bias + sum_ci sum_k act(x)[b,ci,li] w[ci,k,co],  lo = li*stride - pad + k
template <bool XF32, bool YF32>
__global__ __launch_bounds__(256) void convT1d_kernel(ConvParams p) {
    __shared__ float Xs[CI_T][L_T + 8];
    __shared__ __attribute__((aligned(16))) float Ws[CI_T][MAX_K + 2][CO_T];
    const int tid = threadIdx.x;
    const int tc = tid & 15, tl = tid >> 4;
    const int b = blockIdx.z, co0 = blockIdx.y * CO_T, l0 = blockIdx.x * L_T;
    // input positions that can reach outputs [l0, l0+L_T): li in [floor((l0+pad-K+1)/s), floor((l0+L_T-1+pad)/s)]
    int li_lo = l0 + p.pad - p.K + 1;
    li_lo = li_lo >= 0 ? li_lo / p.stride : -((-li_lo + p.stride - 1) / p.stride);
    const int li_hi = (l0 + L_T - 1 + p.pad) / p.stride;
    const int nli = li_hi - li_lo + 1;  // <= L_T/stride + K/stride + 2 <= L_T + 8 for stride >= 1, K <= 2*stride+1

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += CI_T) {
        __syncthreads();
        for (int idx = tid; idx < CI_T * nli; idx += 256) {
            const int c = idx / nli, sp = idx - c * nli;
            const int ci = ci0 + c, li = li_lo + sp;
            float v = 0.f;
            if (ci < p.Cin && li >= 0 && li < p.Lin) {
                v = ld1<XF32>(p.x, ((int64_t)b * p.Cin + ci) * p.Lin + li);
                if (p.act == 1) {
                    float a = p.aa[ci], bb = p.ab[ci];
                    if (p.logscale) { a = __expf(a); bb = __expf(bb); }
                    v = act_apply(v, 1, a, 1.f / (bb + 1e-9f));
                } else if (p.act >= 2) {
                    v = act_apply(v, p.act, p.act_param, 0.f);
                }
            }
            Xs[c][sp] = v;
        }
        for (int idx = tid; idx < CI_T * p.K * CO_T; idx += 256) {
            const int co = idx & (CO_T - 1);
            const int ck = idx >> 6;
            const int c = ck / p.K, k = ck - c * p.K;
            const int ci = ci0 + c;
            float v = 0.f;
            if (ci < p.Cin && co0 + co < p.Cout) v = p.w[((int64_t)ci * p.K + k) * p.Cout + co0 + co];
            Ws[c][k][co] = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lo = l0 + 4 * tl + j;
            const int t = lo + p.pad;
            const int r = t % p.stride;            // first tap
            const int lif = t / p.stride;          // its input position
            for (int k = r, m = 0; k < p.K; k += p.stride, ++m) {
                const int sp = lif - m - li_lo;
                if (sp < 0 || sp >= nli) continue;
#pragma unroll 4
                for (int c = 0; c < CI_T; ++c) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[c][k][4 * tc]);
                    const float xv = Xs[c][sp];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = fmaf(wv[i], xv, acc[i][j]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + 4 * tc + i;
        if (co >= p.Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * tl + j;
            if (l >= p.Lout) continue;
            st1<YF32>(p.y, ((int64_t)b * p.Cout + co) * p.Lout + l, acc[i][j] + bv);
        }
    }
}

// weight norm fold + repack to [Cin][K][Cout]. One workgroup per index of dim 0 of v (the weight_norm dim).
//   conv:  v [Cout][Cin][K], g [Cout]  -> w[ci][k][co] = g[co] v[co][ci][k] / ||v[co]||
//   convT: v [Cin][Cout][K], g [Cin]   -> w[ci][k][co] = g[ci] v[ci][co][k] / ||v[ci]||
__global__ __launch_bounds__(256) void wn_fold_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                      float* __restrict__ w, int d0, int d1, int K, int transposed) {
    __shared__ float red[16];
    const int o = blockIdx.x;
    const int per = d1 * K;
    const float* vp = v + (int64_t)o * per;
    float scale = 1.f;
    if (g) {
        float s = 0.f;
        for (int i = threadIdx.x; i < per; i += 256) s += vp[i] * vp[i];
        s = block_sum(s, red);
        scale = g[o] / sqrtf(s);
    }
    for (int i = threadIdx.x; i < per; i += 256) {
        const int j = i / K, k = i - j * K;
        if (!transposed) {  // o = co, j = ci ; Cout = d0
            w[((int64_t)j * K + k) * d0 + o] = vp[i] * scale;
        } else {            // o = ci, j = co ; Cout = d1
            w[((int64_t)o * K + k) * d1 + j] = vp[i] * scale;
        }
    }
}

template <bool F32>
__global__ __launch_bounds__(256) void snake_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                    const float* __restrict__ alpha, const float* __restrict__ beta,
                                                    int logscale, int C, int L, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / L) % C);
        float a = alpha[c], b = beta[c];
        if (logscale) { a = __expf(a); b = __expf(b); }
        const float xv = ld1<F32>(x, i);
        st1<F32>(y, i, act_apply(xv, 1, a, 1.f / (b + 1e-9f)));
    }
}

// anti-aliased activation (alias-free-torch Activation1d, used by backup/flows.py:266-279,300-313,452-456):
//   2x kaiser-sinc FIR upsample (12 taps, replicate padding) -> snake / snake-beta -> 2x FIR low-pass downsample.
// One workgroup = 256 consecutive outputs of one (batch, channel) row; x segment and the activated 2x-rate
// signal live in LDS, so HBM sees one read and one write per element.
constexpr int A1_T = 256;
template <bool F32>
__global__ __launch_bounds__(256) void act1d_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                    const float* __restrict__ filt, const float* __restrict__ alpha,
                                                    const float* __restrict__ beta, int logscale, int C, int L) {
    __shared__ float xs[A1_T + 16];
    __shared__ float as[2 * A1_T + 16];
    __shared__ float f[12];
    const int row = blockIdx.y;                     // b * C + c
    const int c = row % C;
    const int t0 = blockIdx.x * A1_T;
    const int64_t base = (int64_t)row * L;
    if (threadIdx.x < 12) f[threadIdx.x] = filt[threadIdx.x];
    // up-sampled index range needed: m in [2*t0 - 5, 2*t0 + 2*A1_T + 6]; x_pad index i in [(m+4+1)/2, (m+15)/2]
    const int m0 = 2 * t0 - 5;
    const int i0 = (m0 + 4) / 2 - 1;                // a safe lower bound of x_pad indices used (may be negative)
    for (int j = threadIdx.x; j < A1_T + 16; j += 256) {
        const int xi = min(max(i0 + j - 5, 0), L - 1);   // replicate padding: x_pad[i] = x[clamp(i - 5)]
        xs[j] = ld1<F32>(x, base + xi);
    }
    __syncthreads();
    float a = alpha[c], b = beta[c];
    if (logscale) { a = __expf(a); b = __expf(b); }
    const float inv_b = 1.f / (b + 1e-9f);
    for (int j = threadIdx.x; j < 2 * A1_T + 12; j += 256) {
        // a_pad[m] = act(u[clamp(m - 5, 0, 2L-1)]) with m = 2*t0 + j  ->  u index n:
        const int n = min(max(2 * t0 + j - 5, 0), 2 * L - 1);
        // u[n] = 2 * sum_i x_pad[i] f[n + 15 - 2i],  i = ceil((n+4)/2) .. floor((n+15)/2)
        const int ilo = (n + 5) >> 1;               // ceil((n+4)/2)
        float u = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int i = ilo + q;
            const int tap = n + 15 - 2 * i;
            if (tap >= 0 && tap < 12) u += xs[i - i0] * f[tap];
        }
        u *= 2.f;
        const float sn = sinf(u * a);
        as[j] = u + inv_b * sn * sn;
    }
    __syncthreads();
    const int t = t0 + threadIdx.x;
    if (t < L) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) acc += f[j] * as[2 * threadIdx.x + j];
        st1<F32>(y, base + t, acc);
    }
}

}  // namespace

extern "C" int kalle_act1d_fwd(const void* x, void* y, int dtype, const float* filter12, const float* alpha,
                               const float* beta, int logscale, int B, int C, int L, void* stream) {
    if (!x || !y || !filter12 || !alpha || !beta || B <= 0 || C <= 0 || L <= 0 || (int64_t)B * C > 65535)
        return KALLE_ERR_ARG;
    dim3 grid((L + A1_T - 1) / A1_T, B * C), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KALLE_F32) KALLE_LAUNCH((act1d_kernel<true>), grid, block, 0, st, x, y, filter12, alpha, beta, logscale, C, L);
    else KALLE_LAUNCH((act1d_kernel<false>), grid, block, 0, st, x, y, filter12, alpha, beta, logscale, C, L);
    return kalle_check_launch();
}

extern "C" int kalle_weight_norm_fold(const float* v, const float* g, float* w_packed, int d0, int d1, int ksize,
                                      int transposed, void* stream) {
    if (!v || !w_packed || d0 <= 0 || d1 <= 0 || ksize <= 0) return KALLE_ERR_ARG;
    KALLE_LAUNCH(wn_fold_kernel, dim3(d0), dim3(256), 0, static_cast<hipStream_t>(stream), v, g, w_packed, d0, d1,
                       ksize, transposed);
    return kalle_check_launch();
}

extern "C" int kalle_conv1d_fwd(const void* x, int x_dtype, const float* w_packed, const float* bias,
                                const void* residual, void* y, int y_dtype, int B, int Cin, int Lin, int Cout, int Lout,
                                int ksize, int stride, int padding, int dilation, int act, const float* act_alpha,
                                const float* act_beta, int act_logscale, float act_param, float out_scale, int post,
                                void* stream) {
    if (!x || !w_packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Lin <= 0 || Lout <= 0) return KALLE_ERR_ARG;
    if (ksize <= 0 || ksize > MAX_K || stride <= 0 || dilation <= 0 || padding < 0) return KALLE_ERR_ARG;
    if ((L_T - 1) * stride + (ksize - 1) * dilation + 1 > MAX_SPAN) return KALLE_ERR_UNSUPPORTED;
    // `padding` is the LEFT pad; the right pad is implied by Lout (symmetric, 'same' or causal alike): taps beyond Lin read 0
    if ((int64_t)(Lout - 1) * stride - padding >= Lin) return KALLE_ERR_ARG;
    if (act == 1 && (!act_alpha || !act_beta)) return KALLE_ERR_ARG;
    if (B > 65535 || (Cout + CO_T - 1) / CO_T > 65535) return KALLE_ERR_ARG;
    ConvParams p{x, w_packed, bias, residual, y, B, Cin, Lin, Cout, Lout, ksize, stride, padding, dilation, act, post,
                 act_alpha, act_beta, act_logscale, act_param, out_scale, act == 4 ? 2 * Cin : Cin};
    dim3 grid((Lout + L_T - 1) / L_T, (Cout + CO_T - 1) / CO_T, B), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool xf = x_dtype == KALLE_F32, yf = y_dtype == KALLE_F32;
    if (xf && yf) KALLE_LAUNCH((conv1d_kernel<true, true>), grid, block, 0, st, p);
    else if (xf) KALLE_LAUNCH((conv1d_kernel<true, false>), grid, block, 0, st, p);
    else if (yf) KALLE_LAUNCH((conv1d_kernel<false, true>), grid, block, 0, st, p);
    else KALLE_LAUNCH((conv1d_kernel<false, false>), grid, block, 0, st, p);
    return kalle_check_launch();
}

extern "C" int kalle_conv_transpose1d_fwd(const void* x, int x_dtype, const float* w_packed, const float* bias, void* y,
                                          int y_dtype, int B, int Cin, int Lin, int Cout, int Lout, int ksize,
                                          int stride, int padding, int act, const float* act_alpha,
                                          const float* act_beta, int act_logscale, float act_param, void* stream) {
    if (!x || !w_packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Lin <= 0 || Lout <= 0) return KALLE_ERR_ARG;
    if (ksize <= 0 || ksize > MAX_K + 2 || stride <= 0 || padding < 0) return KALLE_ERR_ARG;
    if (ksize > 2 * stride + 1) return KALLE_ERR_UNSUPPORTED;
    if (Lout > (Lin - 1) * stride - 2 * padding + ksize) return KALLE_ERR_ARG;   // shorter = causal trim of the tail
    if (act == 1 && (!act_alpha || !act_beta)) return KALLE_ERR_ARG;
    if (B > 65535 || (Cout + CO_T - 1) / CO_T > 65535) return KALLE_ERR_ARG;
    if (act == 4) return KALLE_ERR_UNSUPPORTED;
    ConvParams p{x, w_packed, bias, nullptr, y, B, Cin, Lin, Cout, Lout, ksize, stride, padding, 1, act, 0,
                 act_alpha, act_beta, act_logscale, act_param, 1.f, Cin};
    dim3 grid((Lout + L_T - 1) / L_T, (Cout + CO_T - 1) / CO_T, B), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool xf = x_dtype == KALLE_F32, yf = y_dtype == KALLE_F32;
    if (xf && yf) KALLE_LAUNCH((convT1d_kernel<true, true>), grid, block, 0, st, p);
    else if (xf) KALLE_LAUNCH((convT1d_kernel<true, false>), grid, block, 0, st, p);
    else if (yf) KALLE_LAUNCH((convT1d_kernel<false, true>), grid, block, 0, st, p);
    else KALLE_LAUNCH((convT1d_kernel<false, false>), grid, block, 0, st, p);
    return kalle_check_launch();
}

extern "C" int kalle_snake_beta_fwd(const void* x, void* y, int dtype, const float* alpha, const float* beta,
                                    int logscale, int B, int C, int L, void* stream) {
    if (!x || !y || !alpha || !beta || B <= 0 || C <= 0 || L <= 0) return KALLE_ERR_ARG;
    const int64_t total = (int64_t)B * C * L;
    int64_t g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KALLE_F32)
        KALLE_LAUNCH((snake_kernel<true>), dim3((unsigned)g), dim3(256), 0, st, x, y, alpha, beta, logscale, C, L, total);
    else
        KALLE_LAUNCH((snake_kernel<false>), dim3((unsigned)g), dim3(256), 0, st, x, y, alpha, beta, logscale, C, L, total);
    return kalle_check_launch();
}
