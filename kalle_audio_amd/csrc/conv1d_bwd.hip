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
#include <stdlib.h>

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

// ---- weight gradient, LDS-staged (round 3) -----------------------------------------------------------------------------------
// The kernel above reads every V element K times through per-lane global loads and re-applies the input activation (a sine for
// SnakeBeta) on each of them: 28 loads and 28 activations per 112 FMAs - the VAE's backward pass took 17 x its forward.  Here the
// roles follow the forward kernels: a LANE owns one V channel (a wave 64 of them), its accumulators are TU x K partial sums for
// TU U channels, and the reduction walks positions.  Per tile of P positions the workgroup stages the V span it touches
// ((P - 1) stride + (K - 1) dil + 1 positions x 64 channels) into LDS ONCE - coalesced along positions, activated once per
// element, stored position-major with a 65-float row so that both the transposing store and the channel-per-lane reads are
// conflict-free - and its 4 waves (4 x TU U channels) share it.  The U values of a step are wave-uniform: scalar loads through
// the constant address space, fed to v_fma as SGPR operands.  At the end a wave lays each U channel's [64 cv][K] block out in
// LDS exactly as dW stores it and adds it with contiguous 256-byte atomics.
typedef const __attribute__((address_space(4))) float* cfloat_p;

template <int TU, int K>      // K: the tap count itself (1, 4, 7, 8, 16 - what the VAEs use): every tap loop unrolls without a branch
__global__ __launch_bounds__(256) void conv_wgrad_lds_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                             float* __restrict__ dW, int CU, int CV, int MU, int LV, int stride,
                                                             int pad, int dil, int act, const float* __restrict__ alpha,
                                                             const float* __restrict__ beta, int logscale, int chunks_per_b,
                                                             int per_wg, int P, int rows) {
    extern __shared__ float wg_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cv0 = blockIdx.y * 64, cu0 = blockIdx.z * (4 * TU) + wave * TU;
    const int b = blockIdx.x / chunks_per_b, ch = blockIdx.x - b * chunks_per_b;
    const int m_begin = ch * per_wg, m_end = min(m_begin + per_wg, MU);
    float acc[TU][K];
#pragma unroll
    for (int t = 0; t < TU; ++t)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[t][k] = 0.f;
    // row pointers of this wave's TU channels of U (rows past CU are clamped: their sums are never stored)
    cfloat_p up[TU];
#pragma unroll
    for (int t = 0; t < TU; ++t)
        up[t] = reinterpret_cast<cfloat_p>(reinterpret_cast<uintptr_t>(U)) + ((int64_t)b * CU + min(cu0 + t, CU - 1)) * MU;
    for (int m0 = m_begin; m0 < m_end; m0 += P) {
        __syncthreads();                                   // the previous tile has been read
        const int l0 = m0 * stride - pad;
        for (int j = 0; j < 16; ++j) {
            const int cvl = wave * 16 + j, c = cv0 + cvl;
            float a = 0.f, ib = 0.f;
            if (act == 1 && c < CV) {
                a = alpha[c];
                float bb = beta[c];
                if (logscale) { a = __expf(a); bb = __expf(bb); }
                ib = 1.f / (bb + 1e-9f);
            }
            const float* vr = V + ((int64_t)b * CV + min(c, CV - 1)) * LV;
            for (int r = lane; r < rows; r += 64) {
                const int l = l0 + r;
                float t = 0.f;
                if (c < CV && l >= 0 && l < LV) {
                    t = vr[l];
                    if (act) t = act_fwd(t, act, a, ib);
                }
                wg_lds[r * 65 + cvl] = t;
            }
        }
        __syncthreads();
        const int np = min(P, m_end - m0);
        for (int i0 = 0; i0 < np; i0 += 4) {
            float u[TU][4];
            const int m = m0 + i0;
            if (m + 3 < m_end) {                           // (uniform) four positions of every row: scalar loads off one base each
#pragma unroll
                for (int t = 0; t < TU; ++t) {
                    const cfloat_p q4 = up[t] + m;
                    u[t][0] = q4[0]; u[t][1] = q4[1]; u[t][2] = q4[2]; u[t][3] = q4[3];
                }
            } else {                                       // the last, partial block of this workgroup's range
#pragma unroll
                for (int t = 0; t < TU; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[t][q] = m + q < m_end ? up[t][min(m + q, MU - 1)] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* tr = wg_lds + (i0 + q) * stride * 65 + lane;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float v = tr[k * dil * 65];
#pragma unroll
                    for (int t = 0; t < TU; ++t) acc[t][k] = fmaf(u[t][q], v, acc[t][k]);
                }
            }
        }
    }
    __syncthreads();
    float* ep = wg_lds + wave * (64 * K);
#pragma unroll
    for (int t = 0; t < TU; ++t) {
#pragma unroll
        for (int k = 0; k < K; ++k) ep[lane * K + k] = acc[t][k];
        const int cu = cu0 + t;
        if (cu < CU) {
            float* dst = dW + ((int64_t)cu * CV + cv0) * K;
            for (int e = lane; e < 64 * K; e += 64)
                if (cv0 + e / K < CV) atomicAdd(dst + e, ep[e]);
        }
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
    // LDS-staged kernel: the activation (if any) sits on V, V has at least a quarter of a wave's worth of channels
    static const bool lds_off = getenv("KALLE_CONV_WGRAD_V1") != nullptr;       // experiment switch: the per-lane-load kernel
    if (!lds_off && (act_on == 0 || code == 0) && CV >= 16) {
        int P = 64;
        auto rows_of = [&](int p) { return (p - 1) * stride + (ksize - 1) * dilation + 1; };
        while (P > 4 && rows_of(P) * 65 * 4 > 48 * 1024) P >>= 1;
        const int rows = rows_of(P);
        auto launch2 = [&](auto tu_c, auto k_c) {
            constexpr int TU = decltype(tu_c)::value, KT = decltype(k_c)::value;
            const int ty = (CV + 63) / 64, tz = (CU + 4 * TU - 1) / (4 * TU);
            if (ty > 65535 || tz > 65535) return KALLE_ERR_ARG;
            // ~1536 workgroups in all, at least 4 tiles of positions each (the final atomics cost what ~100 positions do)
            int64_t cpb = (1536 + (int64_t)ty * tz * B - 1) / ((int64_t)ty * tz * B);
            const int64_t max_cpb = (MU + 4 * P - 1) / (4 * P);
            cpb = cpb < 1 ? 1 : (cpb > max_cpb ? max_cpb : cpb);
            const int per_wg = (int)(((MU + cpb - 1) / cpb + P - 1) / P * P);
            const int chunks = (MU + per_wg - 1) / per_wg;
            if ((int64_t)chunks * B > 0x7fffffff) return KALLE_ERR_ARG;
            const int lds = std::max(rows * 65 * 4, 4 * 64 * KT * 4);
            KALLE_LAUNCH((conv_wgrad_lds_kernel<TU, KT>), dim3(chunks * B, ty, tz), dim3(256), lds, st, U, V, dW, CU, CV, MU, LV,
                         stride, padding, dilation, code, al, be, logscale, chunks, per_wg, P, rows);
            return kalle_check_launch();
        };
        using I16_ = std::integral_constant<int, 16>;
        using I8_ = std::integral_constant<int, 8>;
        switch (ksize) {
            case 1: return launch2(I16_{}, std::integral_constant<int, 1>{});
            case 4: return launch2(I16_{}, std::integral_constant<int, 4>{});
            case 7: return launch2(I16_{}, std::integral_constant<int, 7>{});
            case 8: return launch2(I16_{}, I8_{});
            case 16: return launch2(I8_{}, I16_{});
            default: break;                  // other tap counts: the per-lane-load kernel below
        }
    }
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
