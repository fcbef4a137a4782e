// Backward of the VAE conv stacks (gfx950): what torch autograd does for the reference when the pretransform is trained
// (`enable_grad`, stable_audio_tools/models/factory.py:77-80; training/diffusion.py:343-346 runs the encoder under
// torch.set_grad_enabled(enable_grad)) - the reference leaves all of it to torch's conv backward.
//
// One fused forward unit of the VAE is  y = conv(act(x)) (+ residual) (-> tanh).  Its backward is split as
//   data gradient    d act(x)  : the FORWARD kernels of conv1d.hip over dy with re-packed weights (a stride-1 conv's is a conv
//                                with flipped taps, a strided conv's a transposed conv, a transposed conv's a strided conv) -
//                                no new kernel, see kalle_weight_norm_fold's flags;
//   weight gradient  dW        : conv_wgrad_kernel below - a position reduction, fp32, no MFMA (north star: conv stacks stay on
//                                the vector ALU);
//   activation       dx, dalpha, dbeta : act_bwd_kernel (SnakeBeta / ELU), per-channel reductions added atomically;
//   bias / tanh                : chan_sum_kernel, tanh_bwd_kernel;
//   weight norm      dg, dv    : wn_bwd_kernel (w = g v / ||v||, one workgroup per slice of dim 0).
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

inline int grid_for(int64_t work_items, int block, int cap = 4096) {
    int64_t g = (work_items + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ float fast_sin(float x) {   // as conv1d.hip: v_sin_f32 takes revolutions; fract() reduces the range
    const float r = x * 0.15915494309189535f;
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
}
__device__ __forceinline__ float act_fwd(float x, int act, float a, float inv_b) {
    if (act == 1) { const float s = fast_sin(x * a); return x + inv_b * s * s; }
    if (act == 2) return x > 0.f ? x : (__expf(x) - 1.f);
    return x;
}

// ---- weight gradient -------------------------------------------------------------------------------------------------------
// dW[cu][cv][k] = sum over (b, m) of U[b, cu, m] * V[b, cv, m*stride - pad + k*dil]        (taps outside V read 0)
//   Conv1d          : U = dy (cu = co), V = act(x) (cv = ci)                -> dW in the module's [Cout][Cin][K] layout
//   ConvTranspose1d : U = act(x) (cu = ci, m = input position), V = dy      -> dW in the module's [Cin][Cout][K] layout
// A lane owns one position m per step and a TU x TV x K block of partial sums; the 4 waves of a workgroup walk different
// position ranges of the same (cu, cv) tile.  Loads are per-lane global loads: consecutive lanes read consecutive positions of
// a row (coalesced, the K taps of neighbouring lanes overlap in L1).  At the end every wave folds its 64 lanes with shuffles
// and adds TU*TV*K values atomically into dW (zeroed by the caller; fp32).
template <int TU, int TV, int KMAX>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                         float* __restrict__ dW, int B, int CU, int CV, int MU, int LV, int K,
                                                         int stride, int pad, int dil, int act_on, int act,
                                                         const float* __restrict__ alpha, const float* __restrict__ beta,
                                                         int logscale, int64_t per_wg) {
    const int cu0 = blockIdx.z * TU, cv0 = blockIdx.y * TV;
    const int lane = threadIdx.x & 63;
    float acc[TU][TV][KMAX];
#pragma unroll
    for (int i = 0; i < TU; ++i)
#pragma unroll
        for (int j = 0; j < TV; ++j)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) acc[i][j][k] = 0.f;
    // activation parameters of the activated operand's channels of this tile
    float pa[TU > TV ? TU : TV], pib[TU > TV ? TU : TV];
    {
        const int n = act_on ? TU : TV, c0 = act_on ? cu0 : cv0, cmax = act_on ? CU : CV;
#pragma unroll
        for (int i = 0; i < (TU > TV ? TU : TV); ++i) {
            pa[i] = 0.f; pib[i] = 0.f;
            if (act == 1 && i < n) {
                const int c = min(c0 + i, cmax - 1);
                float a = alpha[c], bb = beta[c];
                if (logscale) { a = __expf(a); bb = __expf(bb); }
                pa[i] = a; pib[i] = 1.f / (bb + 1e-9f);
            }
        }
    }
    const int64_t total = (int64_t)B * MU;
    const int64_t p0 = (int64_t)blockIdx.x * per_wg, p1 = min(p0 + per_wg, total);
    for (int64_t pp = p0 + threadIdx.x; pp < p1; pp += 256) {
        const int b = (int)(pp / MU), m = (int)(pp - (int64_t)b * MU);
        float u[TU];
#pragma unroll
        for (int i = 0; i < TU; ++i) {
            const int c = cu0 + i;
            float t = c < CU ? U[((int64_t)b * CU + c) * MU + m] : 0.f;
            if (act_on == 1 && act && c < CU) t = act_fwd(t, act, pa[i], pib[i]);
            u[i] = t;
        }
        const int base = m * stride - pad;
#pragma unroll
        for (int j = 0; j < TV; ++j) {
            const int c = cv0 + j;
            const float* vr = V + ((int64_t)b * CV + min(c, CV - 1)) * LV;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < K) {
                    const int li = base + k * dil;
                    float t = (c < CV && li >= 0 && li < LV) ? vr[li] : 0.f;
                    if (act_on == 0 && act && c < CV && li >= 0 && li < LV) t = act_fwd(t, act, pa[j], pib[j]);
#pragma unroll
                    for (int i = 0; i < TU; ++i) acc[i][j][k] = fmaf(u[i], t, acc[i][j][k]);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TU; ++i)
#pragma unroll
        for (int j = 0; j < TV; ++j)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k >= K) continue;                               // (uniform)
                const float s = wave_sum(acc[i][j][k]);
                if (lane == 0 && cu0 + i < CU && cv0 + j < CV)
                    atomicAdd(dW + ((int64_t)(cu0 + i) * CV + (cv0 + j)) * K + k, s);
            }
}

// ---- activation backward: dx = g * act'(x); SnakeBeta also d alpha, d beta (blocks.py:301-339) ----------------------------------
//   y = x + sin^2(a x) / (b + 1e-9), a = e^alpha, b = e^beta (logscale) or the raw parameters:
//   dy/dx = 1 + sin(2 a x) a / (b + 1e-9);  dy/da = sin(2 a x) x / (b + 1e-9);  dy/db = -sin^2(a x) / (b + 1e-9)^2
//   (chain through the exp when logscale: d/dalpha = a dy/da, d/dbeta = b dy/db)
// One workgroup per (b, c) row segment; the two parameter sums are added atomically (dalpha / dbeta pre-zeroed or accumulated).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                      float* __restrict__ dx, int act, const float* __restrict__ alpha,
                                                      const float* __restrict__ beta, int logscale,
                                                      float* __restrict__ dalpha, float* __restrict__ dbeta, int C, int L,
                                                      int seg) {
    __shared__ float red[16];
    const int row = blockIdx.x, c = row % C;          // (rows on grid x: B * C exceeds grid y's 65535 at 2048 channels x B >= 32)
    const int l0 = blockIdx.y * seg, l1 = min(l0 + seg, L);
    const float* xr = x + (int64_t)row * L;
    const float* gr = g + (int64_t)row * L;
    float* dr = dx + (int64_t)row * L;
    float a = 0.f, bb = 1.f, inv_b = 0.f;
    if (act == 1) {
        a = alpha[c]; bb = beta[c];
        if (logscale) { a = __expf(a); bb = __expf(bb); }
        inv_b = 1.f / (bb + 1e-9f);
    }
    float sa = 0.f, sb = 0.f;
    for (int l = l0 + threadIdx.x; l < l1; l += 256) {
        const float xv = xr[l], gv = gr[l];
        float d;
        if (act == 1) {
            const float s = fast_sin(a * xv), s2 = fast_sin(2.f * a * xv);
            d = 1.f + s2 * a * inv_b;
            sa += gv * s2 * xv * inv_b;
            sb -= gv * s * s * inv_b * inv_b;
        } else if (act == 2) {
            d = xv > 0.f ? 1.f : __expf(xv);
        } else {
            d = 1.f;
        }
        dr[l] = gv * d;
    }
    if (act == 1 && dalpha) {
        sa = block_sum(sa, red);
        sb = block_sum(sb, red);
        if (threadIdx.x == 0) {
            atomicAdd(dalpha + c, logscale ? sa * a : sa);
            atomicAdd(dbeta + c, logscale ? sb * bb : sb);
        }
    }
}

// g = dy * (1 - y^2) in place of a tanh at the end of the decoder (autoencoders.py:185)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                       float* __restrict__ g, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        g[i] = dy[i] * (1.f - y[i] * y[i]);
}

// nn.Upsample(scale_factor=s, mode="nearest") along L (autoencoders.py:88): y[r][l] = x[r][l / s]; adjoint: dx[r][m] = sum_j dy[r][m*s+j]
__global__ __launch_bounds__(256) void upsample_nearest_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n_out,
                                                               int Lout, int s) {
    const int Lin = Lout / s;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / Lout;
        const int l = (int)(i - r * Lout);
        y[i] = x[r * Lin + l / s];
    }
}

__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int64_t n_in,
                                                                   int s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (int64_t)gridDim.x * blockDim.x) {
        const float* g = dy + i * s;        // rows are contiguous, so element i of dx owns dy[i*s .. i*s+s)
        float acc = 0.f;
        for (int j = 0; j < s; ++j) acc += g[j];
        dx[i] = acc;
    }
}

// out[c] += sum over (b, l) of x[b, c, l]  (bias gradient)
__global__ __launch_bounds__(256) void chan_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int L,
                                                       int seg) {
    __shared__ float red[16];
    const int row = blockIdx.x, c = row % C;          // (rows on grid x: B * C exceeds grid y's 65535 at 2048 channels x B >= 32)
    const int l0 = blockIdx.y * seg, l1 = min(l0 + seg, L);
    const float* xr = x + (int64_t)row * L;
    float s = 0.f;
    for (int l = l0 + threadIdx.x; l < l1; l += 256) s += xr[l];
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out + c, s);
}

// weight norm backward (torch.nn.utils.weight_norm, dim 0): w = g v / ||v||  per slice o of dim 0 (n = d1 * K elements)
//   dg[o] = <dw, v> / ||v|| ;  dv = g / ||v|| * (dw - v <dw, v> / ||v||^2)
__global__ __launch_bounds__(256) void wn_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                                                     const float* __restrict__ g, float* __restrict__ dv,
                                                     float* __restrict__ dg, int n, int accumulate) {
    __shared__ float red[16];
    const int o = blockIdx.x;
    const float* wp = dw + (int64_t)o * n;
    const float* vp = v + (int64_t)o * n;
    float* op = dv + (int64_t)o * n;
    float nn = 0.f, dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { nn += vp[i] * vp[i]; dot += wp[i] * vp[i]; }
    nn = block_sum(nn, red);
    dot = block_sum(dot, red);
    const float norm = sqrtf(nn), gg = g[o];
    const float s = gg / norm, t = dot / nn;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float r = s * (wp[i] - vp[i] * t);
        op[i] = accumulate ? op[i] + r : r;
    }
    if (threadIdx.x == 0) dg[o] = accumulate ? dg[o] + dot / norm : dot / norm;
}

}  // namespace

extern "C" int kalle_conv_wgrad(const float* U, const float* V, float* dW, int B, int CU, int CV, int MU, int LV, int ksize,
                                int stride, int padding, int dilation, int act_on, const kalle_act* act, void* stream) {
    if (!U || !V || !dW || B <= 0 || CU <= 0 || CV <= 0 || MU <= 0 || LV <= 0) return KALLE_ERR_ARG;
    if (ksize <= 0 || ksize > 16 || stride <= 0 || dilation <= 0 || padding < 0 || (act_on != 0 && act_on != 1)) return KALLE_ERR_ARG;
    int code = 0, logscale = 0;
    const float *al = nullptr, *be = nullptr;
    if (act) {
        code = act->code; logscale = act->logscale; al = act->alpha; be = act->beta;
        if (code < 0 || code > 2 || (code == 1 && (!al || !be))) return KALLE_ERR_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)B * MU;
    // position ranges: enough workgroups to fill the chip a few times over, at least 2048 positions each (the final fold costs
    // TU*TV*K shuffles + atomics per wave)
    auto launch = [&](auto tu_c, auto tv_c, auto km_c) {
        constexpr int TU = decltype(tu_c)::value, TV = decltype(tv_c)::value, KM = decltype(km_c)::value;
        const int ty = (CV + TV - 1) / TV, tz = (CU + TU - 1) / TU;
        if (ty > 65535 || tz > 65535) return KALLE_ERR_ARG;
        int64_t chunks = (2048 + (int64_t)ty * tz - 1) / ((int64_t)ty * tz);         // ~2048 workgroups in all
        const int64_t max_chunks = (total + 2047) / 2048;
        chunks = chunks < 1 ? 1 : (chunks > max_chunks ? max_chunks : chunks);
        const int64_t per_wg = ((total + chunks - 1) / chunks + 255) / 256 * 256;
        const int gx = (int)((total + per_wg - 1) / per_wg);
        KALLE_LAUNCH((conv_wgrad_kernel<TU, TV, KM>), dim3(gx, ty, tz), dim3(256), 0, st, U, V, dW, B, CU, CV, MU, LV, ksize,
                     stride, padding, dilation, act_on, code, al, be, logscale, per_wg);
        return kalle_check_launch();
    };
    using I2 = std::integral_constant<int, 2>;
    using I4 = std::integral_constant<int, 4>;
    using I8 = std::integral_constant<int, 8>;
    using I16 = std::integral_constant<int, 16>;
    if (ksize <= 4) return launch(I4{}, I8{}, I4{});
    if (ksize <= 8) return launch(I4{}, I4{}, I8{});
    return launch(I2{}, I4{}, I16{});
}

extern "C" int kalle_act_bwd(const float* x, const float* g, float* dx, const kalle_act* act, float* dalpha, float* dbeta, int B,
                             int C, int L, void* stream) {
    if (!x || !g || !dx || !act || B <= 0 || C <= 0 || L <= 0 || (int64_t)B * C > 0x7fffffff) return KALLE_ERR_ARG;
    if (act->code < 0 || act->code > 2 || (act->code == 1 && (!act->alpha || !act->beta))) return KALLE_ERR_ARG;
    if (act->code == 1 && ((dalpha == nullptr) != (dbeta == nullptr))) return KALLE_ERR_ARG;
    const int seg = std::max(8192, (L + 65534) / 65535);
    KALLE_LAUNCH(act_bwd_kernel, dim3(B * C, (L + seg - 1) / seg), dim3(256), 0, static_cast<hipStream_t>(stream), x, g, dx,
                 act->code, act->alpha, act->beta, act->logscale, dalpha, dbeta, C, L, seg);
    return kalle_check_launch();
}

extern "C" int kalle_tanh_bwd(const float* dy, const float* y, float* g, int64_t n, void* stream) {
    if (!dy || !y || !g || n <= 0) return KALLE_ERR_ARG;
    KALLE_LAUNCH(tanh_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, g, n);
    return kalle_check_launch();
}

extern "C" int kalle_upsample_nearest(const float* x, float* y, int64_t rows, int L, int scale, int backward, void* stream) {
    if (!x || !y || rows <= 0 || L <= 0 || scale < 1 || scale > 64) return KALLE_ERR_ARG;
    if (backward) {     // x = dy [rows][L*scale] -> y = dx [rows][L]
        const int64_t n = rows * L;
        KALLE_LAUNCH(upsample_nearest_bwd_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n,
                     scale);
    } else {
        const int64_t n = rows * L * scale;
        KALLE_LAUNCH(upsample_nearest_kernel, dim3(grid_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n,
                     L * scale, scale);
    }
    return kalle_check_launch();
}

extern "C" int kalle_channel_sum(const float* x, float* out, int B, int C, int L, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || L <= 0 || (int64_t)B * C > 0x7fffffff) return KALLE_ERR_ARG;
    const int seg = std::max(16384, (L + 65534) / 65535);
    KALLE_LAUNCH(chan_sum_kernel, dim3(B * C, (L + seg - 1) / seg), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, C, L,
                 seg);
    return kalle_check_launch();
}

extern "C" int kalle_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int d0, int n,
                                     int accumulate, void* stream) {
    if (!dw || !v || !g || !dv || !dg || d0 <= 0 || n <= 0) return KALLE_ERR_ARG;
    KALLE_LAUNCH(wn_bwd_kernel, dim3(d0), dim3(256), 0, static_cast<hipStream_t>(stream), dw, v, g, dv, dg, n, accumulate);
    return kalle_check_launch();
}
