// LayerNorm / adaLN / RMSNorm as wavefront reductions (gfx950): one 64-lane wave owns one row, the row
// lives in registers between the statistics pass and the normalise pass, cross-lane sums are shuffles only.
// Reference: stable_audio_tools/models/transformer.py:173-192 (LayerNorm), 658-682 (adaLN modulation),
//            stable_audio_tools/models/blocks.py:211-221,268-299 (RMSNorm / AdaRMSNorm).
#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

// lane `lane` owns, for j < NCH, the 8 columns starting at (j*64 + lane)*8
template <bool F32>
__device__ __forceinline__ void load8(const void* base, int64_t off, float (&v)[8]) {
    if constexpr (F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + off);
        const f32x4 b = *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + off + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
    } else {
        const i32x4 a = *reinterpret_cast<const i32x4*>(static_cast<const bf16_t*>(base) + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = bf16lo((uint32_t)a[i]); v[2 * i + 1] = bf16hi((uint32_t)a[i]); }
    }
}
template <bool F32>
__device__ __forceinline__ void store8(void* base, int64_t off, const float (&v)[8]) {
    if constexpr (F32) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(base) + off) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(static_cast<float*>(base) + off + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
        i32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (int)pack_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<i32x4*>(static_cast<bf16_t*>(base) + off) = o;
    }
}

template <int NCH, bool XF32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int64_t ld_mod, int rpb,
                                                     bf16_t* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
        if (col < D) {
            load8<XF32>(x, (int64_t)row * D + col, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[j][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
        }
    }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
        if (col < D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    const int64_t mb = (int64_t)(row / rpb) * ld_mod;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
        if (col < D) {
            float g[8], o[8];
            load8<true>(gamma, col, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[j][e] - mu) * rs * g[e];
            if (beta) {
                float b[8];
                load8<true>(beta, col, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += b[e];
            }
            if (scale) {
                float sc[8];
                load8<true>(scale, mb + col, sc);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] *= (1.f + sc[e]);
            }
            if (shift) {
                float sh[8];
                load8<true>(shift, mb + col, sh);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] += sh[e];
            }
            store8<false>(y, (int64_t)row * D + col, o);
        }
    }
}

constexpr int LN_BWD_MAX_BLOCKS = 1024;

template <int NCH, bool XF32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ scale,
                                                     int64_t ld_mod, int rpb, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* dres, float* dx,
                                                     bf16_t* __restrict__ dxb,
                                                     float* __restrict__ dgp, float* __restrict__ dbp, int rows, int D,
                                                     int atomic, int ab_mode) {
    // ab_mode 0: the second accumulator is dbeta = sum_rows dy; 1: the column sums of the bf16-rounded dx (the bias gradient of
    // the Linear whose output gradient this dx is - the FF-out bias of the block below - fused here instead of a pass over dxb)
    __shared__ float red[4][64 * 8 + 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float ag[NCH][8], ab[NCH][8], g[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; g[j][e] = 0.f; }
        if (col < D) load8<true>(gamma, col, g[j]);
    }
    const float invD = 1.f / (float)D;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        const int64_t mb = (int64_t)(row / rpb) * ld_mod;
        float xh[NCH][8], dh[NCH][8], rv[NCH][8];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * 8;
            if (col < D) {
                float xv[8], d[8];
                load8<XF32>(x, (int64_t)row * D + col, xv);
                load8<false>(dy, (int64_t)row * D + col, d);
                // the residual-stream gradient travels with x and dy (issued after the reductions it would expose a second
                // load latency per row: it may alias dx, so the compiler cannot move it up by itself)
                if (dres) load8<true>(dres, (int64_t)row * D + col, rv[j]);
                if (scale) {
                    float sc[8];
                    load8<true>(scale, mb + col, sc);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] *= (1.f + sc[e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[j][e] = (xv[e] - mu) * rs;
                    ag[j][e] += d[e] * xh[j][e];
                    if (ab_mode == 0) ab[j][e] += d[e];
                    dh[j][e] = d[e] * g[j][e];
                    c1 += dh[j][e];
                    c2 += dh[j][e] * xh[j][e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[j][e] = 0.f; dh[j][e] = 0.f; }
            }
        }
        c1 = wave_sum(c1) * invD;
        c2 = wave_sum(c2) * invD;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * 8;
            if (col < D) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (dh[j][e] - c1 - xh[j][e] * c2);
                if (dres) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += rv[j][e];
                }
                store8<true>(dx, (int64_t)row * D + col, o);
                if (dxb) store8<false>(dxb, (int64_t)row * D + col, o);
                if (ab_mode == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ab[j][e] += bf16_to_f32(f32_to_bf16(o[e]));
                }
            }
        }
    }
    // combine the 4 waves' partial dgamma/dbeta through LDS; one partial row per block
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = pass == 0 ? ag[j][e] : ab[j][e];
            __syncthreads();
            float* outp = pass == 0 ? dgp : dbp;
            if (outp) {
                for (int i = threadIdx.x; i < 512; i += 256) {
                    const int col = j * 512 + i;
                    if (col < D) {
                        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
                        if (atomic) atomicAdd(outp + col, v);      // straight into the [D] accumulator: no reduction pass
                        else outp[(int64_t)blockIdx.x * D + col] = v;
                    }
                }
            }
        }
    }
}

// dscale[b,d] = sum_t dy * (xhat*gamma+beta) ; dshift[b,d] = sum_t dy    (one thread per (b, d-pair))
template <bool XF32>
__global__ void adaln_mod_bwd_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     float* __restrict__ dscale, float* __restrict__ dshift, int64_t ld_mod, int rpb,
                                     int D) {
    const int b = blockIdx.y;
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (col >= D) return;
    const float g0 = gamma[col], g1 = gamma[col + 1];
    const float b0 = beta ? beta[col] : 0.f, b1 = beta ? beta[col + 1] : 0.f;
    float s0 = 0.f, s1 = 0.f, h0 = 0.f, h1 = 0.f;
    for (int t = 0; t < rpb; ++t) {
        const int64_t row = (int64_t)b * rpb + t;
        const float mu = mean[row], rs = rstd[row];
        const uint32_t dw = *reinterpret_cast<const uint32_t*>(dy + row * D + col);
        const float d0 = bf16lo(dw), d1 = bf16hi(dw);
        float x0, x1;
        if constexpr (XF32) {
            const f32x2 xv = *reinterpret_cast<const f32x2*>(static_cast<const float*>(x) + row * D + col);
            x0 = xv[0]; x1 = xv[1];
        } else {
            const uint32_t xw = *reinterpret_cast<const uint32_t*>(static_cast<const bf16_t*>(x) + row * D + col);
            x0 = bf16lo(xw); x1 = bf16hi(xw);
        }
        s0 += d0 * ((x0 - mu) * rs * g0 + b0);
        s1 += d1 * ((x1 - mu) * rs * g1 + b1);
        h0 += d0; h1 += d1;
    }
    dscale[(int64_t)b * ld_mod + col] = s0; dscale[(int64_t)b * ld_mod + col + 1] = s1;
    dshift[(int64_t)b * ld_mod + col] = h0; dshift[(int64_t)b * ld_mod + col + 1] = h1;
}

// ---- RMSNorm ---------------------------------------------------------------------------------
template <int NCH, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void rms_fwd_kernel(const void* __restrict__ x, const float* __restrict__ scale,
                                                      int64_t ld_scale, int rpb, void* __restrict__ y,
                                                      float* __restrict__ rrms, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[NCH][8];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
        if (col < D) {
            load8<XF32>(x, (int64_t)row * D + col, v[j]);
#pragma unroll
            for (int e = 0; e < 8; ++e) q += v[j][e] * v[j][e];
        }
    }
    const float rr = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0 && rrms) rrms[row] = rr;
    const int64_t sb = (int64_t)(row / rpb) * ld_scale;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int col = (j * 64 + lane) * 8;
        if (col < D) {
            float s[8], o[8];
            load8<true>(scale, sb + col, s);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[j][e] * (s[e] * rr);
            store8<YF32>(y, (int64_t)row * D + col, o);
        }
    }
}

template <int NCH, bool XF32, bool DYF32>
__global__ __launch_bounds__(256) void rms_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                      const float* __restrict__ scale, int64_t ld_scale, int rpb,
                                                      const float* __restrict__ rrms, float* __restrict__ dx,
                                                      float* __restrict__ dsp, const float* __restrict__ dres,
                                                      bf16_t* __restrict__ dxb, int rows, int D, int atomic) {
    __shared__ float red[4][64 * 8 + 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float as[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) as[j][e] = 0.f;
    const float invD = 1.f / (float)D;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const float rr = rrms[row];
        const int64_t sb = (int64_t)(row / rpb) * ld_scale;
        float xv[NCH][8], dh[NCH][8], rv[NCH][8];
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * 8;
            if (col < D) {
                float d[8], s[8];
                load8<XF32>(x, (int64_t)row * D + col, xv[j]);
                load8<DYF32>(dy, (int64_t)row * D + col, d);
                load8<true>(scale, sb + col, s);
                if (dres) load8<true>(dres, (int64_t)row * D + col, rv[j]);   // (with x and dy: one latency per row)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    as[j][e] += d[e] * xv[j][e] * rr;
                    dh[j][e] = d[e] * s[e];
                    c += dh[j][e] * xv[j][e];
                }
            }
        }
        c = wave_sum(c) * invD * rr * rr * rr;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int col = (j * 64 + lane) * 8;
            if (col < D) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = dh[j][e] * rr - xv[j][e] * c;
                if (dres) {   // fused residual-stream gradient add
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += rv[j][e];
                }
                store8<true>(dx, (int64_t)row * D + col, o);
                if (dxb) store8<false>(dxb, (int64_t)row * D + col, o);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = as[j][e];
        __syncthreads();
        if (dsp) {
            for (int i = threadIdx.x; i < 512; i += 256) {
                const int col = j * 512 + i;
                if (col < D) {
                    const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
                    if (atomic) atomicAdd(dsp + col, v);
                    else dsp[(int64_t)blockIdx.x * D + col] = v;
                }
            }
        }
    }
}

// ---- column sums --------------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ in, int64_t ld, float* __restrict__ out,
                                                     int rows, int cols, int rows_per_slab) {
    __shared__ float red[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 128 + lane * 2;
    const int r0 = blockIdx.y * rows_per_slab;
    const int r1 = min(rows, r0 + rows_per_slab);
    float s0 = 0.f, s1 = 0.f;
    if (col < cols) {
        for (int r = r0 + wave; r < r1; r += 4) {
            if constexpr (F32) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(static_cast<const float*>(in) + (int64_t)r * ld + col);
                s0 += v[0]; s1 += v[1];
            } else {
                const uint32_t w = *reinterpret_cast<const uint32_t*>(static_cast<const bf16_t*>(in) + (int64_t)r * ld + col);
                s0 += bf16lo(w); s1 += bf16hi(w);
            }
        }
    }
    red[wave][lane * 2] = s0;
    red[wave][lane * 2 + 1] = s1;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int c = blockIdx.x * 128 + threadIdx.x;
        if (c < cols) {
            const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            atomicAdd(out + c, t);
        }
    }
}

#define DISPATCH_NCH(D, CALL)                    \
    do {                                         \
        const int nch_ = ((D) + 511) / 512;      \
        if (nch_ <= 1) { CALL(1); }              \
        else if (nch_ <= 2) { CALL(2); }         \
        else if (nch_ <= 3) { CALL(3); }         \
        else if (nch_ <= 4) { CALL(4); }         \
        else if (nch_ <= 6) { CALL(6); }         \
        else { CALL(8); }                        \
    } while (0)


// ---- per-head normalisation of q / k before the rotary embedding (transformer.py:422-428, `qk_norm`):
//   mode 1 "l2": F.normalize(x, dim=-1) = x / max(||x||, 1e-12);   mode 2 "ln": LayerNorm(64, eps 1e-6) with gamma / beta.
// 8 lanes own one 64-wide head (8 bf16 each, one 16-byte load), sums are 3 xor-shuffles inside the 8-lane group.
__device__ __forceinline__ float group8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

__global__ __launch_bounds__(256) void head_norm_fwd_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y,
                                                            int64_t ldy, float* __restrict__ stat, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int mode, int64_t rows, int heads) {
    const int sub = threadIdx.x & 7;
    const int64_t nunit = rows * heads;
    float gm[8], bt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = 1.f; bt[e] = 0.f; }
    if (mode == 2) {
        load8<true>(gamma, sub * 8, gm);
        if (beta) load8<true>(beta, sub * 8, bt);
    }
    for (int64_t u = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); u < nunit; u += (int64_t)gridDim.x * 32) {
        const int64_t row = u / heads;
        const int h = (int)(u - row * heads);
        float v[8], o[8];
        load8<false>(x, row * ldx + h * 64 + sub * 8, v);
        float s = 0.f, mu = 0.f, rs;
        if (mode == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e] * v[e];
            const float nrm = sqrtf(group8_sum(s));
            rs = 1.f / fmaxf(nrm, 1e-12f);
            mu = nrm > 1e-12f ? 0.f : 1.f;          // 1: the clamp is active, the map is linear (x * 1e12)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = v[e] * rs;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += v[e];
            mu = group8_sum(s) * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q += d * d; }
            rs = rsqrtf(group8_sum(q) * (1.f / 64.f) + 1e-6f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[e] - mu) * rs * gm[e] + bt[e];
        }
        store8<false>(y, row * ldy + h * 64 + sub * 8, o);
        if (sub == 0) { stat[2 * u] = mu; stat[2 * u + 1] = rs; }
    }
}

__global__ __launch_bounds__(256) void head_norm_bwd_kernel(const bf16_t* __restrict__ x, int64_t ldx, const float* __restrict__ stat,
                                                            const bf16_t* __restrict__ g, int64_t ldg, bf16_t* __restrict__ dx,
                                                            int64_t lddx, const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int mode, int64_t rows, int heads) {
    __shared__ float red[4][2][64];
    const int sub = threadIdx.x & 7;
    const int64_t nunit = rows * heads;
    float gm[8], ag[8], ab[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gm[e] = 1.f; ag[e] = 0.f; ab[e] = 0.f; }
    if (mode == 2) load8<true>(gamma, sub * 8, gm);
    for (int64_t u = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); u < nunit; u += (int64_t)gridDim.x * 32) {
        const int64_t row = u / heads;
        const int h = (int)(u - row * heads);
        float v[8], gv[8], o[8];
        load8<false>(x, row * ldx + h * 64 + sub * 8, v);
        load8<false>(g, row * ldg + h * 64 + sub * 8, gv);
        const float mu = stat[2 * u], rs = stat[2 * u + 1];
        if (mode == 1) {
            float d = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) d += v[e] * rs * gv[e];
            d = mu != 0.f ? 0.f : group8_sum(d);        // (every lane of the group takes the same branch)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs * (gv[e] - v[e] * rs * d);
        } else {
            float s1 = 0.f, s2 = 0.f, xh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[e] = (v[e] - mu) * rs;
                const float gh = gv[e] * gm[e];
                s1 += gh;
                s2 += gh * xh[e];
                ag[e] += gv[e] * xh[e];
                ab[e] += gv[e];
            }
            s1 = group8_sum(s1) * (1.f / 64.f);
            s2 = group8_sum(s2) * (1.f / 64.f);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rs * (gv[e] * gm[e] - s1 - xh[e] * s2);
        }
        store8<false>(dx, row * lddx + h * 64 + sub * 8, o);
    }
    if (mode != 2 || !dgamma) return;
    // column sums of this workgroup: across the 8 groups of a wave by shuffles, across the 4 waves through LDS, one atomic per column
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float a = ag[e], b = ab[e];
        a += __shfl_xor(a, 8); a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 8); b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (lane < 8) { red[wave][0][lane * 8 + e] = a; red[wave][1][lane * 8 + e] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, c = threadIdx.x & 63;
        const float t = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        float* dst = which ? dbeta : dgamma;
        if (dst) atomicAdd(dst + c, t);
    }
}

}  // namespace

extern "C" int kalle_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                                   const float* scale, const float* shift, int64_t ld_mod, int rows_per_batch,
                                   void* y, float* mean, float* rstd, int rows, int D, float eps, void* stream) {
    if (!x || !gamma || !y || rows <= 0 || D <= 0 || (D & 7) || D > 4096) return KALLE_ERR_ARG;
    if ((scale || shift) && ((ld_mod & 3) || rows_per_batch <= 0)) return KALLE_ERR_ARG;
    const int rpb = rows_per_batch > 0 ? rows_per_batch : 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4), block(256);
#define CALL(N)                                                                                                    \
    if (x_dtype == KALLE_F32)                                                                                      \
        KALLE_LAUNCH((ln_fwd_kernel<N, true>), grid, block, 0, st, x, gamma, beta, scale, shift, ld_mod, rpb, \
                           static_cast<bf16_t*>(y), mean, rstd, rows, D, eps);                                     \
    else                                                                                                           \
        KALLE_LAUNCH((ln_fwd_kernel<N, false>), grid, block, 0, st, x, gamma, beta, scale, shift, ld_mod,    \
                           rpb, static_cast<bf16_t*>(y), mean, rstd, rows, D, eps);
    DISPATCH_NCH(D, CALL);
#undef CALL
    return kalle_check_launch();
}

extern "C" int kalle_layernorm_bwd_parts(int rows) {
    // rows per workgroup (4 waves = 4 rows per pass).  Every workgroup ends with 2 D atomics (or a partial row) whatever it did, so
    // below ~8000 rows - where the 1024-workgroup cap does not bind - 8 rows per workgroup instead of 4 halve that tail: the B = 16
    // train step (2016 rows, 72 launches) 31.4 -> 30.9 ms, 16 rows 31.1 (too few waves in flight); KALLE_LN_BWD_RPB to experiment
    static const int rpb = getenv("KALLE_LN_BWD_RPB") ? atoi(getenv("KALLE_LN_BWD_RPB")) : 8;
    const int b = (rows + rpb - 1) / (rpb > 0 ? rpb : 4);
    return b < LN_BWD_MAX_BLOCKS ? (b > 0 ? b : 1) : LN_BWD_MAX_BLOCKS;
}

static int ln_bwd_launch(const void* dy, const void* x, int x_dtype, const float* gamma, const float* scale,
                         int64_t ld_mod, int rows_per_batch, const float* mean, const float* rstd, const float* dres,
                         float* dx_out, void* dx_bf16, float* dgamma_part, float* dbeta_part, int rows, int D, int atomic,
                         void* stream, int ab_mode = 0);

extern "C" int kalle_layernorm_bwd(const void* dy, const void* x, int x_dtype, const float* gamma,
                                   const float* scale, int64_t ld_mod, int rows_per_batch, const float* mean,
                                   const float* rstd, const float* dres, float* dx_out, void* dx_bf16,
                                   float* dgamma_part,
                                   float* dbeta_part, int rows, int D, void* stream) {
    return ln_bwd_launch(dy, x, x_dtype, gamma, scale, ld_mod, rows_per_batch, mean, rstd, dres, dx_out, dx_bf16,
                         dgamma_part, dbeta_part, rows, D, 0, stream);
}

extern "C" int kalle_layernorm_bwd_acc(const void* dy, const void* x, int x_dtype, const float* gamma,
                                       const float* scale, int64_t ld_mod, int rows_per_batch, const float* mean,
                                       const float* rstd, const float* dres, float* dx_out, void* dx_bf16,
                                       float* dgamma_acc, float* dbeta_acc, int rows, int D, void* stream) {
    return ln_bwd_launch(dy, x, x_dtype, gamma, scale, ld_mod, rows_per_batch, mean, rstd, dres, dx_out, dx_bf16,
                         dgamma_acc, dbeta_acc, rows, D, 1, stream);
}

extern "C" int kalle_layernorm_bwd_colsum(const void* dy, const void* x, int x_dtype, const float* gamma,
                                          const float* scale, int64_t ld_mod, int rows_per_batch, const float* mean,
                                          const float* rstd, const float* dres, float* dx_out, void* dx_bf16,
                                          float* dgamma_acc, float* dx_colsum_acc, int rows, int D, void* stream) {
    if (!dx_colsum_acc) return KALLE_ERR_ARG;
    return ln_bwd_launch(dy, x, x_dtype, gamma, scale, ld_mod, rows_per_batch, mean, rstd, dres, dx_out, dx_bf16,
                         dgamma_acc, dx_colsum_acc, rows, D, 1, stream, 1);
}

static int ln_bwd_launch(const void* dy, const void* x, int x_dtype, const float* gamma, const float* scale,
                         int64_t ld_mod, int rows_per_batch, const float* mean, const float* rstd, const float* dres,
                         float* dx_out, void* dx_bf16, float* dgamma_part, float* dbeta_part, int rows, int D, int atomic,
                         void* stream, int ab_mode) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx_out || rows <= 0 || (D & 7) || D <= 0 || D > 4096)
        return KALLE_ERR_ARG;
    const int rpb = rows_per_batch > 0 ? rows_per_batch : 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(kalle_layernorm_bwd_parts(rows)), block(256);
#define CALL(N)                                                                                                   \
    if (x_dtype == KALLE_F32)                                                                                     \
        KALLE_LAUNCH((ln_bwd_kernel<N, true>), grid, block, 0, st, static_cast<const bf16_t*>(dy), x, gamma, \
                           scale, ld_mod, rpb, mean, rstd, dres, dx_out, static_cast<bf16_t*>(dx_bf16), dgamma_part,   \
                           dbeta_part, rows, D, atomic, ab_mode);                                                  \
    else                                                                                                          \
        KALLE_LAUNCH((ln_bwd_kernel<N, false>), grid, block, 0, st, static_cast<const bf16_t*>(dy), x,      \
                           gamma, scale, ld_mod, rpb, mean, rstd, dres, dx_out, static_cast<bf16_t*>(dx_bf16),         \
                           dgamma_part, dbeta_part, rows, D, atomic, ab_mode);
    DISPATCH_NCH(D, CALL);
#undef CALL
    return kalle_check_launch();
}

extern "C" int kalle_adaln_mod_bwd(const void* dy, const void* x, int x_dtype, const float* gamma, const float* beta,
                                   const float* mean, const float* rstd, float* dscale, float* dshift,
                                   int64_t ld_mod, int nbatch, int rows_per_batch, int D, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dscale || !dshift || nbatch <= 0 || rows_per_batch <= 0 || (D & 7))
        return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 block(128), grid((D / 2 + 127) / 128, nbatch);
    if (x_dtype == KALLE_F32)
        KALLE_LAUNCH((adaln_mod_bwd_kernel<true>), grid, block, 0, st, static_cast<const bf16_t*>(dy), x, gamma,
                           beta, mean, rstd, dscale, dshift, ld_mod, rows_per_batch, D);
    else
        KALLE_LAUNCH((adaln_mod_bwd_kernel<false>), grid, block, 0, st, static_cast<const bf16_t*>(dy), x,
                           gamma, beta, mean, rstd, dscale, dshift, ld_mod, rows_per_batch, D);
    return kalle_check_launch();
}

extern "C" int kalle_rmsnorm_fwd(const void* x, int x_dtype, const float* scale, int64_t ld_scale, int rows_per_batch,
                                 void* y, int y_dtype, float* rrms, int rows, int D, float eps, void* stream) {
    if (!x || !scale || !y || rows <= 0 || D <= 0 || (D & 7) || D > 4096) return KALLE_ERR_ARG;
    const int rpb = rows_per_batch > 0 ? rows_per_batch : rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((rows + 3) / 4), block(256);
#define CALL(N)                                                                                                    \
    if (x_dtype == KALLE_F32 && y_dtype == KALLE_F32)                                                              \
        KALLE_LAUNCH((rms_fwd_kernel<N, true, true>), grid, block, 0, st, x, scale, ld_scale, rpb, y, rrms,  \
                           rows, D, eps);                                                                          \
    else if (x_dtype == KALLE_F32)                                                                                 \
        KALLE_LAUNCH((rms_fwd_kernel<N, true, false>), grid, block, 0, st, x, scale, ld_scale, rpb, y, rrms, \
                           rows, D, eps);                                                                          \
    else if (y_dtype == KALLE_F32)                                                                                 \
        KALLE_LAUNCH((rms_fwd_kernel<N, false, true>), grid, block, 0, st, x, scale, ld_scale, rpb, y, rrms, \
                           rows, D, eps);                                                                          \
    else                                                                                                           \
        KALLE_LAUNCH((rms_fwd_kernel<N, false, false>), grid, block, 0, st, x, scale, ld_scale, rpb, y,      \
                           rrms, rows, D, eps);
    DISPATCH_NCH(D, CALL);
#undef CALL
    return kalle_check_launch();
}

static int rms_bwd_launch(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* scale, int64_t ld_scale,
                          int rows_per_batch, const float* rrms, float* dx, float* dscale_part, const float* dres,
                          void* dx_bf16, int rows, int D, int atomic, void* stream);
extern "C" int kalle_rmsnorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* scale,
                                 int64_t ld_scale, int rows_per_batch, const float* rrms, float* dx,
                                 float* dscale_part, const float* dres, void* dx_bf16, int rows, int D,
                                 void* stream) {
    return rms_bwd_launch(dy, dy_dtype, x, x_dtype, scale, ld_scale, rows_per_batch, rrms, dx, dscale_part, dres, dx_bf16,
                          rows, D, 0, stream);
}
extern "C" int kalle_rmsnorm_bwd_acc(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* scale,
                                     int64_t ld_scale, int rows_per_batch, const float* rrms, float* dx,
                                     float* dscale_acc, const float* dres, void* dx_bf16, int rows, int D,
                                     void* stream) {
    return rms_bwd_launch(dy, dy_dtype, x, x_dtype, scale, ld_scale, rows_per_batch, rrms, dx, dscale_acc, dres, dx_bf16,
                          rows, D, 1, stream);
}
static int rms_bwd_launch(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* scale, int64_t ld_scale,
                          int rows_per_batch, const float* rrms, float* dx, float* dscale_part, const float* dres,
                          void* dx_bf16, int rows, int D, int atomic, void* stream) {
    if (!dy || !x || !scale || !rrms || !dx || rows <= 0 || D <= 0 || (D & 7) || D > 4096) return KALLE_ERR_ARG;
    const int rpb = rows_per_batch > 0 ? rows_per_batch : rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(kalle_layernorm_bwd_parts(rows)), block(256);
#define CALL(N)                                                                                                     \
    if (x_dtype == KALLE_F32 && dy_dtype == KALLE_F32)                                                              \
        KALLE_LAUNCH((rms_bwd_kernel<N, true, true>), grid, block, 0, st, dy, x, scale, ld_scale, rpb, rrms,  \
                           dx, dscale_part, dres, static_cast<bf16_t*>(dx_bf16), rows, D, atomic);                          \
    else if (x_dtype == KALLE_F32)                                                                                  \
        KALLE_LAUNCH((rms_bwd_kernel<N, true, false>), grid, block, 0, st, dy, x, scale, ld_scale, rpb, rrms, \
                           dx, dscale_part, dres, static_cast<bf16_t*>(dx_bf16), rows, D, atomic);                          \
    else if (dy_dtype == KALLE_F32)                                                                                 \
        KALLE_LAUNCH((rms_bwd_kernel<N, false, true>), grid, block, 0, st, dy, x, scale, ld_scale, rpb, rrms, \
                           dx, dscale_part, dres, static_cast<bf16_t*>(dx_bf16), rows, D, atomic);                          \
    else                                                                                                            \
        KALLE_LAUNCH((rms_bwd_kernel<N, false, false>), grid, block, 0, st, dy, x, scale, ld_scale, rpb,      \
                           rrms, dx, dscale_part, dres, static_cast<bf16_t*>(dx_bf16), rows, D, atomic);
    DISPATCH_NCH(D, CALL);
#undef CALL
    return kalle_check_launch();
}

extern "C" int kalle_colsum(const void* in, int in_dtype, int64_t ld, float* out, int rows, int cols, int accumulate,
                            void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0 || (cols & 1) || (ld & 1)) return KALLE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!accumulate) {
        if (hipMemsetAsync(out, 0, sizeof(float) * cols, st) != hipSuccess) return KALLE_ERR_LAUNCH;
    }
    const int gx = (cols + 127) / 128;
    int slabs = (2048 + gx - 1) / gx;
    if (slabs > (rows + 15) / 16) slabs = (rows + 15) / 16;
    if (slabs < 1) slabs = 1;
    const int rps = (rows + slabs - 1) / slabs;
    slabs = (rows + rps - 1) / rps;
    dim3 grid(gx, slabs), block(256);
    if (in_dtype == KALLE_F32)
        KALLE_LAUNCH((colsum_kernel<true>), grid, block, 0, st, in, ld, out, rows, cols, rps);
    else
        KALLE_LAUNCH((colsum_kernel<false>), grid, block, 0, st, in, ld, out, rows, cols, rps);
    return kalle_check_launch();
}

extern "C" int kalle_head_norm_fwd(const void* x, int64_t ldx, int64_t x_off, void* y, int64_t ldy, int64_t y_off, float* stat,
                                   const float* gamma, const float* beta, int mode, int64_t rows, int heads, void* stream) {
    if (!x || !y || !stat || rows <= 0 || heads <= 0 || (mode != 1 && mode != 2) || (mode == 2 && !gamma)) return KALLE_ERR_ARG;
    if ((ldx & 7) || (ldy & 7) || (x_off & 7) || (y_off & 7) || ldx < (int64_t)heads * 64 || ldy < (int64_t)heads * 64 || x_off < 0 ||
        y_off < 0)
        return KALLE_ERR_ARG;
    const int64_t nunit = rows * heads;
    const int grid = (int)((nunit + 31) / 32 < 4096 ? (nunit + 31) / 32 : 4096);
    KALLE_LAUNCH(head_norm_fwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                 static_cast<const bf16_t*>(x) + x_off, ldx, static_cast<bf16_t*>(y) + y_off, ldy, stat, gamma, beta, mode, rows, heads);
    return kalle_check_launch();
}

extern "C" int kalle_head_norm_bwd(const void* x, int64_t ldx, int64_t x_off, const float* stat, const void* g, int64_t ldg,
                                   int64_t g_off, void* dx, int64_t lddx, int64_t dx_off, const float* gamma, float* dgamma,
                                   float* dbeta, int mode, int64_t rows, int heads, void* stream) {
    if (!x || !g || !dx || !stat || rows <= 0 || heads <= 0 || (mode != 1 && mode != 2) || (mode == 2 && !gamma)) return KALLE_ERR_ARG;
    if ((ldx & 7) || (ldg & 7) || (lddx & 7) || (x_off & 7) || (g_off & 7) || (dx_off & 7) || x_off < 0 || g_off < 0 || dx_off < 0)
        return KALLE_ERR_ARG;
    if (ldx < (int64_t)heads * 64 || ldg < (int64_t)heads * 64 || lddx < (int64_t)heads * 64) return KALLE_ERR_ARG;
    const int64_t nunit = rows * heads;
    const int grid = (int)((nunit + 31) / 32 < 2048 ? (nunit + 31) / 32 : 2048);
    KALLE_LAUNCH(head_norm_bwd_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                 static_cast<const bf16_t*>(x) + x_off, ldx, stat, static_cast<const bf16_t*>(g) + g_off, ldg,
                 static_cast<bf16_t*>(dx) + dx_off, lddx, gamma, dgamma, dbeta, mode, rows, heads);
    return kalle_check_launch();
}
