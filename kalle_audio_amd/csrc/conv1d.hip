// 1-D convolution stacks of the audio VAEs (gfx950, vector ALUs only - no MFMA, per the north-star spec):
// weight-norm fold, input activation (SnakeBeta / ELU / LeakyReLU / WaveNet gate) fused into the input staging,
// dilated / strided / causal conv and transposed conv, and (conv + bias + residual) * scale (+= y) -> activation -> tanh
// fused into the store.
// Reference: stable_audio_tools/models/autoencoders.py:39-62 (ResidualUnit), 64-81 (EncoderBlock), 83-114
// (DecoderBlock), 116-191 (Oobleck encoder/decoder); blocks.py:301-339 (SnakeBeta); dac.nn.layers.WNConv1d ->
// torch.nn.utils.weight_norm (w = g * v / ||v||, norm over all dims but 0); backup/flows.py (mel-VAE).
//
// Layout: activations (B, C, L) row-major (L contiguous, as torch), fp32 or bf16; weights are repacked once per
// forward by kalle_weight_norm_fold into [Cin][K][CoutP] fp32 (CoutP = Cout rounded up to 8).
//
// Main kernels ("v2", stride-1 conv and every transposed conv): a wave owns 8 output channels for 64*LPT positions;
// lane l handles positions l, l+64, ... so every LDS read of the input span is lane-consecutive (conflict-free for any
// dilation / alignment); the 8 weights of a (ci, tap) are wave-uniform, so they are fetched with SCALAR loads straight
// from the packed array (scalar cache / L2) and feed v_pk_fma_f32 as SGPR operands - no LDS traffic for weights.  The
// input span is staged through registers (global loads of chunk n+1 in flight while chunk n is computed), activated when
// it is written to LDS, double-buffered, one barrier per 8-channel chunk; the (channel, tap) loop is software-pipelined by
// hand (loads of step t+1 issued right after the wait for step t).  A transposed conv of stride S is S independent
// ceil(K/S)-tap convolutions (one per output phase) run back to back by the same workgroup.
// Fallback kernels ("v1": strided conv, gated input, mixed dtypes): 64 co x 64 positions per workgroup, 4x4 per thread.
#include <cstdlib>
#include "common.h"
#include "../../include/kalle_hip.h"

namespace {

constexpr int CO_T = 64, L_T = 64, CI_T = 8;
constexpr int MAX_SPAN = 576;   // (L_T-1)*stride + (K-1)*dil + 1 must fit
constexpr int MAX_K = 16;

__device__ __forceinline__ float fast_sin(float x) {   // v_sin_f32 takes revolutions; fract() does the range reduction
    const float r = x * 0.15915494309189535f;
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(r));
}
// act: 1 snake(-beta) x + sin^2(a x) * inv_b, 2 ELU, 3 LeakyReLU(a)
__device__ __forceinline__ float act_apply(float x, int act, float a, float inv_b) {
    if (act == 1) {
        const float s = fast_sin(x * a);
        return x + inv_b * s * s;
    }
    if (act == 2) return x > 0.f ? x : (__expf(x) - 1.f);
    if (act == 3) return x > 0.f ? x : x * a;
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
    int B, Cin, Lin, Cout, Lout, K, stride, pad, dil, act, post;   // post bits: 1 tanh, 2 accumulate into y
    const float* aa; const float* ab; int logscale;
    float act_param;
    float out_scale;
    int CoutP;  // row stride of the packed weights = Cout rounded up to 8 (pad columns are zero)
    int xC;     // channels of the x tensor (2*Cin for the gated activation, else Cin)
    int pact; const float* paa; const float* pab; int plogscale; float pparam;   // activation applied to the output
    void* y_raw;               // dual output: the un-activated value also goes here (same shape / dtype as y)
    int nco, ntile, co_fast;   // v2 grid: blockIdx.x enumerates (position tile, channel tile), channel tile fastest if co_fast
};

// (conv + bias + residual) * out_scale (+= y) -> post activation -> tanh -> store
template <bool XF32, bool YF32, bool RES = true>
__device__ __forceinline__ void conv_store(const ConvParams& p, int co, int64_t oi, float v, float bv, float pa,
                                           float pinv_b) {
    v += bv;
    if (RES && p.res) v += ld1<XF32>(p.res, oi);
    v *= p.out_scale;
    if (p.post & 2) v += ld1<YF32>(p.y, oi);
    if (p.y_raw) st1<YF32>(p.y_raw, oi, v);         // consumers that need the raw value (residual paths)
    if (p.pact) v = act_apply(v, p.pact, pa, pinv_b);
    if (p.post & 1) v = tanhf(v);
    st1<YF32>(p.y, oi, v);
}
__device__ __forceinline__ void post_act_params(const ConvParams& p, int co, float& pa, float& pinv_b) {
    pa = p.pparam;
    pinv_b = 0.f;
    if (p.pact == 1) {
        pa = p.paa[co];
        float bb = p.pab[co];
        if (p.plogscale) { pa = __expf(pa); bb = __expf(bb); }
        pinv_b = 1.f / (bb + 1e-9f);
    }
}

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
            if (ci < p.Cin && co0 + co < p.Cout) v = p.w[((int64_t)ci * p.K + k) * p.CoutP + co0 + co];
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
        float pa, pinv_b;
        post_act_params(p, co, pa, pinv_b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * tl + j;
            if (l >= p.Lout) continue;
            conv_store<XF32, YF32>(p, co, ((int64_t)b * p.Cout + co) * p.Lout + l, acc[i][j], bv, pa, pinv_b);
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
            if (ci < p.Cin && co0 + co < p.Cout) v = p.w[((int64_t)ci * p.K + k) * p.CoutP + co0 + co];
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
        float pa, pinv_b;
        post_act_params(p, co, pa, pinv_b);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int l = l0 + 4 * tl + j;
            if (l >= p.Lout) continue;
            conv_store<XF32, YF32>(p, co, ((int64_t)b * p.Cout + co) * p.Lout + l, acc[i][j], bv, pa, pinv_b);
        }
    }
}



// ------------------------------------------------------------------------------------------------ v2 core
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float* cfloat_p;   // constant address space: uniform loads become SMEM

struct ConvPass {        // one pass of the core = one output phase
    int in0;             // x index of span slot 0
    int span;            // slots to stage (<= SPAN)
    int xs0;             // span slot read by tap 0 at local position 0
    int ntaps;           // taps per input channel (K, or the taps of this phase of a transposed conv)
    int S;               // input phases of a strided conv (1 otherwise): x index i is staged at LDS slot
                         // (i - in0) % S * pspan + (i - in0) / S, so a tap reads lane-consecutive slots at any stride
    int lgS, pspan, ph0; // log2(S), slots per phase row, phase of tap 0
    int xtap;            // slot advance per tap inside a phase run (S > 1: pspan = next phase)
    int xtap_wrap;       // slot advance when the phase wraps (S == 1: every tap - the dilation, or -1)
    int64_t w0;          // first weight row of ci = 0, in rows of CoutP
    int64_t wtap, wchan; // weight row advance per tap / from the last tap of ci to the first tap of ci + 1
    int64_t o0;          // output position of local position 0
    int ostride;         // output positions per local position (1, or the stride of a transposed conv)
};

// COW output channels per wave, 64*LPT positions per wave; WCO of the NW waves are spread over output channels, the other
// NW/WCO over positions (tiny-Cout layers: WCO = 1); CI input channels per staged chunk; SPAN = LDS row length.
// NW = 8 (512 threads) stages x once for 128 output channels: half the x traffic and half the staging instructions per
// FMA of the 4-wave tiling - what the HBM-side pointwise (k = 1) convs need.
template <int COW, int LPT, int WCO, int CI, int SPAN, bool XF32, bool YF32, int NW = 4>
__device__ __forceinline__ void conv_core(const ConvParams& p, const ConvPass& g, float (*Xs)[SPAN], int b, int co_w,
                                          int lane, int wave, int wave_l) {
    constexpr int LW = 64 * LPT;
    constexpr int NS = SPAN / 64;                     // span slots per lane
    constexpr int NH = CI / NW;                       // channels of a chunk staged by one wave
    const int nchunks = (p.Cin + CI - 1) / CI;

    f32x2 acc[COW][LPT / 2];                          // packed along positions: (l_2j, l_2j+1) share one v_pk_fma_f32
#pragma unroll
    for (int i = 0; i < COW; ++i)
#pragma unroll
        for (int j = 0; j < LPT / 2; ++j) acc[i][j] = f32x2{0.f, 0.f};
    // the residual is loaded straight into the accumulators up front (all loads in flight together, their latency hidden
    // behind the first chunk) - in the store loop every load would sit behind the previous store (they may alias)
    if (p.res) {                                      // clamped addresses, no branches: invalid lanes are never stored
#pragma unroll
        for (int i = 0; i < COW; ++i) {
            const int64_t rb = ((int64_t)b * p.Cout + min(co_w + i, p.Cout - 1)) * p.Lout;
#pragma unroll
            for (int j = 0; j < LPT / 2; ++j) {
                const int64_t la = g.o0 + (int64_t)(wave_l * LW + lane + 128 * j) * g.ostride;
                const int64_t lb = la + 64 * g.ostride;
                const float ra = ld1<XF32>(p.res, rb + min(max(la, (int64_t)0), (int64_t)p.Lout - 1));
                const float rc = ld1<XF32>(p.res, rb + min(max(lb, (int64_t)0), (int64_t)p.Lout - 1));
                acc[i][j] = f32x2{ra, rc};
            }
        }
    }

    float stg[NH][NS];                                // wave w stages channels w, w+NW, ... of a chunk
    float s_a[NH], s_b[NH];                           // their activation parameters (scalar loads, issued with the data)
    const cfloat_p aa_c = reinterpret_cast<cfloat_p>(reinterpret_cast<uintptr_t>(p.aa));
    const cfloat_p ab_c = reinterpret_cast<cfloat_p>(reinterpret_cast<uintptr_t>(p.ab));
    auto gload = [&](int ci0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int ci = ci0 + wave + NW * h;
            if (p.act == 1) {
                s_a[h] = aa_c[min(ci, p.Cin - 1)];
                s_b[h] = ab_c[min(ci, p.Cin - 1)];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int sp = lane + 64 * s, li = g.in0 + sp;
                float v = 0.f;
                if (sp < g.span && ci < p.Cin && li >= 0 && li < p.Lin)
                    v = ld1<XF32>(p.x, ((int64_t)b * p.xC + ci) * p.Lin + li);
                stg[h][s] = v;
            }
        }
    };
    auto lstore = [&](int ci0, int buf) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int c = wave + NW * h, ci = ci0 + c;
            float a = p.act_param, inv_b = 0.f;
            if (p.act == 1) {
                a = s_a[h];
                float bb = s_b[h];
                if (p.logscale) { a = __expf(a); bb = __expf(bb); }
                inv_b = 1.f / (bb + 1e-9f);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int sp = lane + 64 * s, li = g.in0 + sp;
                if (sp < g.span) {
                    float v = stg[h][s];
                    if (p.act && ci < p.Cin && li >= 0 && li < p.Lin) v = act_apply(v, p.act, a, inv_b);
                    const int slot = g.S == 1 ? sp : (sp & (g.S - 1)) * g.pspan + (sp >> g.lgS);
                    Xs[buf * CI + c][slot] = v;
                }
            }
        }
    };
    // A wave whose channel group lies wholly beyond the packed row (Cout = 16 under a 32-channel workgroup: waves 2, 3) still
    // runs the loop for the barriers and the staging; its weight reads are redirected to column 0 - read at co_w they would
    // run up to COW floats past the END of the array on the last (ci, tap) row (a fault when the array ends a mapped segment).
    // (A group that is partly valid needs CoutP % COW == 0: pick_tile only offers COW = 16 then.)
    const int co_ld = co_w + COW <= p.CoutP ? co_w : 0;
    const cfloat_p wbase = reinterpret_cast<cfloat_p>(reinterpret_cast<uintptr_t>(p.w)) + co_ld;
    const cfloat_p wlast = wbase + ((int64_t)p.Cin * p.K - 1) * p.CoutP;
    const int nwrap = (g.ph0 + g.ntaps - 1) / g.S;   // phase wraps among the ntaps-1 advances of one channel
    const int xwrap = SPAN - ((g.ntaps - 1 - nwrap) * g.xtap + nwrap * g.xtap_wrap);
    const int64_t wtap = g.wtap * p.CoutP, wchan = g.wchan * p.CoutP;

    // The scalar weight loads only run one step ahead, so they must hit in L2: every thread touches one 64-B line of the
    // NEXT chunk's weight rows with a vector load (same vmcnt batch as the x loads), which pulls the rows of all four
    // waves from HBM / MALL into this XCD's L2 a whole chunk before they are needed.
    constexpr int LPR = WCO * COW * 4 / 64 > 0 ? WCO * COW * 4 / 64 : 1;   // 64-B lines per weight row of this workgroup
    float wt0 = 0.f, wt1 = 0.f;                       // touched values: consumed (= waited for) only at the next lstore
    auto wprefetch = [&](int ci0) {
        const int per_c = g.ntaps * LPR;
        const int total = min(CI, p.Cin - ci0) * per_c;
        const int co_g = co_w - (wave % WCO) * COW;
        auto touch = [&](int idx) {
            idx = idx < total ? idx : 0;              // clamped, branch-free: a redundant touch of the first line
            const int c = idx / per_c, rem = idx - c * per_c;
            const int m = rem / LPR;
            int col = co_g + 16 * (rem - m * LPR);
            col = col < p.CoutP ? col : co_g;
            const int64_t row = min(g.w0 + (int64_t)(ci0 + c) * p.K + (int64_t)m * g.wtap, (int64_t)p.Cin * p.K - 1);
            return p.w[row * p.CoutP + col];
        };
        if (total > 0) {
            wt0 = touch(threadIdx.x);
            wt1 = touch(threadIdx.x + 64 * NW);
        }
    };
    auto consume_touch = [&]() { asm volatile("" ::"v"(wt0), "v"(wt1)); };

    wprefetch(0);
    gload(0);
    consume_touch();
    if (nchunks > 1) wprefetch(CI);
    lstore(0, 0);
    __syncthreads();
    for (int n = 0; n < nchunks; ++n) {
        const int ci0 = n * CI, buf = n & 1;
        if (n + 1 < nchunks) {
            consume_touch();                             // (loads issued a whole chunk ago)
            if (n + 2 < nchunks) wprefetch(ci0 + 2 * CI);
            gload(ci0 + CI);
        }
        const int cmax = min(CI, p.Cin - ci0);
        // flattened (channel, tap) loop, software-pipelined by hand: the loads of step t+1 are issued right after the
        // wait for step t's operands, so LDS + scalar-cache latency hides behind the packed FMAs of step t.
        // (scalar loads return out of order, so any wait on them is lgkmcnt(0): the wait must precede the prefetch.)
        const int T = cmax * g.ntaps;
        cfloat_p wr = wbase + (g.w0 + (int64_t)ci0 * p.K) * p.CoutP;
        if (wr > wlast) wr = wlast;
        const float* xs = &Xs[buf * CI][g.xs0 + wave_l * LW + lane];
        int kk = 0, ph = g.ph0;
        f32x2 w0[COW / 2], w1[COW / 2];
        f32x2 x0[LPT / 2], x1[LPT / 2];
        auto ld = [&](f32x2 (&w)[COW / 2], f32x2 (&x)[LPT / 2]) {
#pragma unroll
            for (int i = 0; i < COW / 2; ++i) w[i] = f32x2{wr[2 * i], wr[2 * i + 1]};
#pragma unroll
            for (int j = 0; j < LPT / 2; ++j) x[j] = f32x2{xs[128 * j], xs[128 * j + 64]};
            if (++kk == g.ntaps) {
                kk = 0; ph = g.ph0; xs += xwrap; wr += wchan;
            } else {
                wr += wtap;
                if (++ph == g.S) { ph = 0; xs += g.xtap_wrap; } else xs += g.xtap;
            }
            if (wr > wlast) wr = wlast;              // the one-step-ahead prefetch never leaves the array
        };
        // acc(l_2j, l_2j+1) += w_i * (x_2j, x_2j+1): the weight is one half of an SGPR pair, broadcast by op_sel (hipcc
        // would copy odd weights into even SGPRs first, which makes it wait on the prefetch it has just issued)
        auto fma = [&](const f32x2 (&w)[COW / 2], const f32x2 (&x)[LPT / 2]) {
#pragma unroll
            for (int j = 0; j < LPT / 2; ++j)
#pragma unroll
                for (int i = 0; i < COW / 2; ++i) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[2 * i][j]) : "s"(w[i]), "v"(x[j]));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                        : "+v"(acc[2 * i + 1][j]) : "s"(w[i]), "v"(x[j]));
                }
        };
        auto wait_lgkm = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
            __builtin_amdgcn_sched_barrier(0);
        };
        if (T > 0) {
            ld(w0, x0);
            int t = 0;
            for (; t + 2 <= T; t += 2) {
                wait_lgkm();
                ld(w1, x1);
                fma(w0, x0);
                wait_lgkm();
                ld(w0, x0);                          // step t+2 (may be one past the chunk: unused)
                fma(w1, x1);
            }
            if (t < T) fma(w0, x0);
        }
        if (n + 1 < nchunks) lstore(ci0 + CI, buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < COW; ++i) {
        const int co = co_w + i;
        if (co >= p.Cout) continue;
        const float bv = p.bias ? p.bias[co] : 0.f;
        float pa, pinv_b;
        post_act_params(p, co, pa, pinv_b);
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const int64_t l = g.o0 + (int64_t)(wave_l * LW + lane + 64 * j) * g.ostride;
            if (l < 0 || l >= p.Lout) continue;
            conv_store<XF32, YF32, false>(p, co, ((int64_t)b * p.Cout + co) * p.Lout + l, acc[i][j >> 1][j & 1], bv, pa,
                                          pinv_b);
        }
    }
}

template <int COW, int LPT, int WCO, int CI, int SPAN, bool XF32, bool YF32, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv1d_v2_kernel(ConvParams p) {
    constexpr int LT = 64 * LPT * (NW / WCO);
    __shared__ float Xs[2 * CI + 1][SPAN];           // two buffers + one pad row for the one-step-ahead prefetch
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bt = p.co_fast ? blockIdx.x / p.nco : blockIdx.x % p.ntile;
    const int bc = p.co_fast ? blockIdx.x % p.nco : blockIdx.x / p.ntile;
    const int l0 = bt * LT;
    ConvPass g;
    g.ntaps = p.K;
    g.w0 = 0;
    g.wtap = 1;
    g.wchan = 1;
    g.o0 = l0;
    g.ostride = 1;
    if (p.stride == 1) {
        g.S = 1; g.lgS = 0; g.pspan = 0; g.ph0 = 0;
        g.in0 = l0 - p.pad;
        g.span = (LT - 1) + (p.K - 1) * p.dil + 1;
        g.xs0 = 0;
        g.xtap = 0;
        g.xtap_wrap = p.dil;
    } else {                                          // power-of-two stride, dilation 1 (host-checked)
        const int S = p.stride;
        const int padq = (p.pad + S - 1) / S;         // in0 is a multiple of S: tap k of local position l sits at slot l*S + k + d
        const int d = padq * S - p.pad;
        g.S = S; g.lgS = 31 - __builtin_clz(S);
        g.pspan = LT + 64 / S;                        // (pspan mod 64) = 64/S: the de-interleaving LDS writes are conflict-free
        g.ph0 = d;
        g.in0 = (l0 - padq) * S;
        g.span = (LT - 1) * S + p.K + d;
        g.xs0 = d * g.pspan;
        g.xtap = g.pspan;
        g.xtap_wrap = 1 - (S - 1) * g.pspan;
    }
    conv_core<COW, LPT, WCO, CI, SPAN, XF32, YF32, NW>(p, g, Xs, blockIdx.z, bc * (WCO * COW) + (wave % WCO) * COW, lane,
                                                            wave, wave / WCO);
}

// transposed conv: output lo = q*S + r - pad (phase r < S, input position q):
//   y[q*S + r - pad] = sum_ci sum_m x[q - m] w[ci, r + m*S, co],  m < ceil((K - r)/S)
// i.e. S independent ceil(K/S)-tap convolutions whose outputs interleave.  One workgroup = one phase of one
// (64*LPT*(4/WCO) input positions q) x (channel tile).  Workgroup ids are laid out so that the S phases of a tile differ
// by multiples of 8: the hardware deals consecutive ids round-robin to the 8 XCDs, so all phases of a tile run on the
// SAME XCD within a few ids of each other and their interleaved 4-byte stores merge in that XCD's L2 before write-back.
template <int COW, int LPT, int WCO, int CI, int SPAN, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void convT1d_v2_kernel(ConvParams p) {
    constexpr int LT = 64 * LPT * (4 / WCO);
    __shared__ float Xs[2 * CI + 1][SPAN];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int S = p.stride;
    const int n = blockIdx.x;
    const int r = (n >> 3) % S;
    const int tile = (n / (8 * S)) * 8 + (n & 7);    // (position tile, channel tile) pair, position tiles fastest
    if (tile >= p.ntile * p.nco) return;
    const int bt = tile % p.ntile, bc = tile / p.ntile;
    const int q0 = bt * LT;
    const int mmax = (p.K + S - 1) / S;
    ConvPass g;
    g.ntaps = r < p.K ? (p.K - r + S - 1) / S : 0;
    g.in0 = q0 - (mmax - 1);
    g.span = LT + mmax - 1;
    g.xs0 = mmax - 1;
    g.S = 1; g.lgS = 0; g.pspan = 0; g.ph0 = 0;
    g.xtap = 0;
    g.xtap_wrap = -1;
    g.w0 = r;
    g.wtap = S;
    g.wchan = p.K - (int64_t)(g.ntaps - 1) * S;
    g.o0 = (int64_t)q0 * S + r - p.pad;
    g.ostride = S;
    conv_core<COW, LPT, WCO, CI, SPAN, XF32, YF32>(p, g, Xs, blockIdx.z, bc * (WCO * COW) + (wave % WCO) * COW, lane,
                                                        wave, wave / WCO);
}

// ------------------------------------------------------------------------------------------------ few positions, many channels
// The top of the VAE (C = 512 .. 2048 at a few hundred to a few thousand positions - all of a single-clip decode) does not
// have enough positions to fill 256 CUs with the position-per-lane tiling above.  Here the roles are swapped: a LANE owns 4
// output channels (a wave 256 of them) and 16 consecutive positions live in its accumulators; the weights of a (ci, tap)
// step are then a coalesced 16-byte-per-lane VECTOR load from the packed [Cin][K][CoutP] array (prefetched 4 steps ahead,
// counted vmcnt), and the 16 input values of the step are wave-uniform: ONE scalar load from a zero-padded, already
// activated copy of x (kalle_conv_pad_act), fed to v_pk_fma_f32 as SGPR pairs.  No LDS, no barriers, waves independent.
// phases > 1 (strided conv): padded index j is stored at (j % phases) * (Lp / phases) + j / phases, so the inputs of a tap
// for consecutive outputs are consecutive
__global__ __launch_bounds__(256) void pad_act_kernel(const float* __restrict__ x, float* __restrict__ xp, int C, int Lin,
                                                      int Lp, int pad, int act, const float* __restrict__ aa,
                                                      const float* __restrict__ ab, int logscale, float act_param,
                                                      int phases, int gx) {
    const int row = blockIdx.x / gx;                // b * C + c (rows on grid x: B * C exceeds grid y's 65535 at 1024 channels x 64 chunks)
    const int bx = blockIdx.x - row * gx;
    const int c = row % C;
    float a = act_param, inv_b = 0.f;
    if (act == 1) {
        a = aa[c];
        float bb = ab[c];
        if (logscale) { a = __expf(a); bb = __expf(bb); }
        inv_b = 1.f / (bb + 1e-9f);
    }
    for (int j = bx * 256 + threadIdx.x; j < Lp; j += gx * 256) {
        const int li = j - pad;
        float v = 0.f;
        if (li >= 0 && li < Lin) {
            v = x[(int64_t)row * Lin + li];
            if (act) v = act_apply(v, act, a, inv_b);
        }
        const int slot = phases == 1 ? j : (j % phases) * (Lp / phases) + j / phases;
        xp[(int64_t)row * Lp + slot] = v;
    }
}

struct ConvCParams {
    const float* xp; const float* w; const float* bias; const float* res; float* y; float* y_raw;
    int B, Cin, Lp, Cout, CoutP, Lout, K, dil, post;
    int split;           // 1, 2 or 4 waves of a workgroup share one position tile and split the input channels
    int ks; float* part; // ks > 1: blockIdx.z = b * ks + slice - workgroups split the input channels too; each writes its raw
                         // partial sums to part[slice][b][co][l] and cfirst_finish_kernel applies bias / residual / epilogue
    // geometry of one pass: conv (nphase 1, taps K, weight row k, x offset k * dil, output l) or one phase r of a transposed
    // conv (taps ceil((K - r) / S), weight row r + m * S, x offset -m, output q * S + r - pad)
    int nphase, npos, xtap, ostride, opad, xlead;
    int xS, xLq, xph0;   // strided conv: input phases, phase-row length, phase of tap 0 (xS = 1 otherwise)
    float out_scale;
    int pact; const float* paa; const float* pab; int plogscale; float pparam;
};

__global__ __launch_bounds__(256) void conv1d_cfirst_kernel(ConvCParams p) {
    constexpr int P = 16, D = 4;                    // positions per wave, weight prefetch depth
    __shared__ float red[3 * 64 * 64];              // partial accumulators of the input-channel splits (48 KiB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.z / p.ks, ksl = blockIdx.z - b * p.ks;   // batch item, workgroup-level input-channel slice
    const int S = p.split;                          // waves per position tile
    const int sp = wave % S;                        // this wave's input-channel slice (within the workgroup's)
    const int ph = blockIdx.x % p.nphase;           // output phase (transposed conv), 0 otherwise
    const int l0 = ((blockIdx.x / p.nphase) * (4 / S) + wave / S) * P;   // first (input-side) position of the tile
    const bool live = l0 < p.npos;                  // (whole tile groups stay together: sp-waves of a tile share `live`)
    const int co = blockIdx.y * 256 + 4 * lane;
    const int cw = min(co, p.CoutP - 4);            // clamped weight column (lanes past Cout are never stored)
    const int cper = (p.Cin + S * p.ks - 1) / (S * p.ks);
    const int ci_lo = min((ksl * S + sp) * cper, p.Cin), ci_hi = min(ci_lo + cper, p.Cin);
    const int kt = p.nphase == 1 ? p.K : (ph < p.K ? (p.K - ph + p.nphase - 1) / p.nphase : 0);   // taps of this pass
    const int wstep = p.nphase;                     // weight k advance per tap (1 for a conv)
    const int T = live ? (ci_hi - ci_lo) * kt : 0;
    const int64_t oo = p.nphase == 1 ? 0 : ph - p.opad;   // output index of position q: q * ostride + oo

    f32x2 acc[4][P / 2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < P / 2; ++j) acc[c][j] = f32x2{0.f, 0.f};
    if (p.res && sp == 0 && live && p.ks == 1) {    // residual straight into the accumulators (clamped, branch-free)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* rp = p.res + ((int64_t)b * p.Cout + min(co + c, p.Cout - 1)) * p.Lout;
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {
                const int64_t la = (int64_t)(l0 + 2 * j) * p.ostride + oo, lb = la + p.ostride;
                acc[c][j] = f32x2{rp[min(max(la, (int64_t)0), (int64_t)p.Lout - 1)],
                                  rp[min(max(lb, (int64_t)0), (int64_t)p.Lout - 1)]};
            }
        }
    }

    // weight row of step t = (ci, m): ci * K + ph + m * wstep;  x row offset: ci * Lp + xlead + m * xtap.  Both are walked
    // with running scalar pointers (adds only - recomputing them from (ci, m) costs ~40 scalar instructions per step)
    const cfloat_p xb = reinterpret_cast<cfloat_p>(reinterpret_cast<uintptr_t>(p.xp)) + (int64_t)b * p.Cin * p.Lp + l0;
    int64_t xoff = (int64_t)ci_lo * p.Lp + p.xlead;
    const int nwr = p.xS > 1 ? (p.xph0 + kt - 1) / p.xS : 0;          // phase wraps among a channel's kt-1 tap advances
    const int64_t xphw = 1 - (int64_t)(p.xS - 1) * p.xLq;
    const int64_t xwrap = (int64_t)p.Lp - (p.xS > 1 ? (int64_t)(kt - 1 - nwr) * p.xLq + nwr * xphw : (int64_t)(kt - 1) * p.xtap);
    int k_n = 0, ph_x = p.xph0;
    auto xoff_next = [&]() {
        const int64_t o = xoff;
        if (++k_n == kt) { k_n = 0; ph_x = p.xph0; xoff += xwrap; }
        else if (p.xS > 1) { if (++ph_x == p.xS) { ph_x = 0; xoff += xphw; } else xoff += p.xLq; }
        else xoff += p.xtap;
        return o;
    };
    const float* wrow = p.w + ((int64_t)ci_lo * p.K + min(ph, p.K - 1)) * p.CoutP;      // uniform
    const float* const wlast = p.w + ((int64_t)p.Cin * p.K - 1) * p.CoutP;
    const int64_t wtap = (int64_t)wstep * p.CoutP, wwrap = ((int64_t)p.K - (int64_t)(kt - 1) * wstep) * p.CoutP;
    int k_w = 0;
    auto wload = [&]() {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + cw);
        if (++k_w >= kt) { k_w = 0; wrow += wwrap; } else wrow += wtap;
        if (wrow > wlast) wrow = wlast;              // prefetch past the last step: clamped, unused
        return v;
    };
    f32x4 wq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) wq[d] = wload();
    f32x2 xa[P / 2], xn[P / 2];
    auto xload = [&](f32x2 (&xv)[P / 2], int64_t off) {
#pragma unroll
        for (int j = 0; j < P / 2; ++j) xv[j] = f32x2{xb[off + 2 * j], xb[off + 2 * j + 1]};
    };
    auto fma = [&](const f32x4& wv, const f32x2 (&xv)[P / 2]) {
        const f32x2 w01 = f32x2{wv[0], wv[1]}, w23 = f32x2{wv[2], wv[3]};
#pragma unroll
        for (int j = 0; j < P / 2; ++j) {
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[0][j]) : "v"(w01), "s"(xv[j]));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[1][j]) : "v"(w01), "s"(xv[j]));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[2][j]) : "v"(w23), "s"(xv[j]));
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[3][j]) : "v"(w23), "s"(xv[j]));
        }
    };
    if (T > 0) xload(xa, xoff_next());
    // steps in groups of D so the prefetch slots are static registers; x double-buffered by name (two steps per pair)
    auto step = [&](int t, int slot, f32x2 (&xc)[P / 2], f32x2 (&xnext)[P / 2]) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): x of this step is here
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < T) xload(xnext, xoff_next());
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");        // weights of this step (3 younger loads may fly)
        const f32x4 wv = wq[slot];
        wq[slot] = wload();
        fma(wv, xc);
    };
    int t = 0;
#pragma unroll 1
    for (; t + 4 <= T; t += 4) {
        step(t, 0, xa, xn);
        step(t + 1, 1, xn, xa);
        step(t + 2, 2, xa, xn);
        step(t + 3, 3, xn, xa);
    }
    if (t < T) { step(t, 0, xa, xn); ++t; }
    if (t < T) { step(t, 1, xn, xa); ++t; }
    if (t < T) { step(t, 2, xa, xn); ++t; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (S > 1) {                                    // fold the input-channel slices: slice 0 of each tile collects
        const int tile = wave / S;
        if (sp > 0) {
            float* rp = red + ((tile * (S - 1) + sp - 1) * 64) * 64 + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < P; ++j) rp[(c * P + j) * 64] = acc[c][j >> 1][j & 1];
        }
        __syncthreads();
        if (sp > 0) return;
        for (int q = 0; q < S - 1; ++q) {
            const float* rp = red + ((tile * (S - 1) + q) * 64) * 64 + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < P; ++j) acc[c][j >> 1][j & 1] += rp[(c * P + j) * 64];
        }
    }
    if (!live) return;
    if (p.ks > 1) {                                 // raw partial sums; the epilogue runs in cfirst_finish_kernel
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int cc = co + c;
            if (cc >= p.Cout) continue;
            float* pp = p.part + (((int64_t)ksl * p.B + b) * p.Cout + cc) * p.Lout;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const int64_t l = (int64_t)(l0 + j) * p.ostride + oo;
                if (l >= 0 && l < p.Lout) pp[l] = acc[c][j >> 1][j & 1];
            }
        }
        return;
    }

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cc = co + c;
        if (cc >= p.Cout) continue;
        const float bv = p.bias ? p.bias[cc] : 0.f;
        float pa = p.pparam, pinv_b = 0.f;
        if (p.pact == 1) {
            pa = p.paa[cc];
            float bb = p.pab[cc];
            if (p.plogscale) { pa = __expf(pa); bb = __expf(bb); }
            pinv_b = 1.f / (bb + 1e-9f);
        }
        float* yp = p.y + ((int64_t)b * p.Cout + cc) * p.Lout;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int64_t l = (int64_t)(l0 + j) * p.ostride + oo;
            if (l < 0 || l >= p.Lout) continue;
            float v = (acc[c][j >> 1][j & 1] + bv) * p.out_scale;   // (residual already inside acc)
            if (p.post & 2) v += yp[l];
            if (p.y_raw) p.y_raw[((int64_t)b * p.Cout + cc) * p.Lout + l] = v;
            if (p.pact) v = act_apply(v, p.pact, pa, pinv_b);
            if (p.post & 1) v = tanhf(v);
            yp[l] = v;
        }
    }
}

// sums the input-channel slices of a split conv1d_cfirst launch and applies the conv epilogue (same order as above)
__global__ __launch_bounds__(256) void cfirst_finish_kernel(ConvCParams p) {
    const int64_t n = (int64_t)p.B * p.Cout * p.Lout;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int cc = (int)((i / p.Lout) % p.Cout);
        float v = p.res ? p.res[i] : 0.f;
        for (int s = 0; s < p.ks; ++s) v += p.part[(int64_t)s * n + i];
        v = (v + (p.bias ? p.bias[cc] : 0.f)) * p.out_scale;
        if (p.post & 2) v += p.y[i];
        if (p.y_raw) p.y_raw[i] = v;
        if (p.pact) {
            float pa = p.pparam, pinv_b = 0.f;
            if (p.pact == 1) {
                pa = p.paa[cc];
                float bb = p.pab[cc];
                if (p.plogscale) { pa = __expf(pa); bb = __expf(bb); }
                pinv_b = 1.f / (bb + 1e-9f);
            }
            v = act_apply(v, p.pact, pa, pinv_b);
        }
        if (p.post & 1) v = tanhf(v);
        p.y[i] = v;
    }
}

// weight norm fold + repack to [Cin][K][CoutP] (CoutP = Cout rounded up to 8, pad columns zeroed). One workgroup per index of dim 0 of v (the weight_norm dim).
//   conv:  v [Cout][Cin][K], g [Cout]  -> w[ci][k][co] = g[co] v[co][ci][k] / ||v[co]||
//   convT: v [Cin][Cout][K], g [Cin]   -> w[ci][k][co] = g[ci] v[ci][co][k] / ||v[ci]||
__global__ __launch_bounds__(256) void wn_fold_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                      float* __restrict__ w, int d0, int d1, int K, int flags) {
    __shared__ float red[16];
    const int transposed = flags & 1, flip = flags & 2;     // flip: tap k stored at K-1-k (the data gradient of a stride-1 conv)
    const int o = blockIdx.x;
    const int per = d1 * K;
    const int cout = transposed ? d1 : d0, coutp = (cout + 7) & ~7;
    if (!transposed && o >= d0) {       // pad output channel: zero column
        for (int i = threadIdx.x; i < per; i += 256) w[(int64_t)i * coutp + o] = 0.f;   // i = j*K + k
        return;
    }
    const float* vp = v + (int64_t)o * per;
    float scale = 1.f;
    if (g) {
        float s = 0.f;
        for (int i = threadIdx.x; i < per; i += 256) s += vp[i] * vp[i];
        s = block_sum(s, red);
        scale = g[o] / sqrtf(s);
    }
    for (int i = threadIdx.x; i < per; i += 256) {
        const int j = i / K, k0 = i - j * K, k = flip ? K - 1 - k0 : k0;
        if (!transposed) {  // o = co, j = ci ; Cout = d0
            w[((int64_t)j * K + k) * coutp + o] = vp[i] * scale;
        } else {            // o = ci, j = co ; Cout = d1
            w[((int64_t)o * K + k) * coutp + j] = vp[i] * scale;
        }
    }
    if (transposed)
        for (int i = threadIdx.x; i < (coutp - d1) * K; i += 256) {
            const int k = i / (coutp - d1), j = d1 + i - k * (coutp - d1);
            w[((int64_t)o * K + k) * coutp + j] = 0.f;
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
        const float sn = sinf(xv * a);            // accurate sine: this kernel is HBM-bound anyway
        st1<F32>(y, i, xv + sn * sn / (b + 1e-9f));
    }
}

// anti-aliased activation (alias-free-torch Activation1d, used by backup/flows.py:266-279,300-313,452-456):
//   2x kaiser-sinc FIR upsample (12 taps, replicate padding) -> snake / snake-beta -> 2x FIR low-pass downsample.
// One workgroup = 256 consecutive outputs of one (batch, channel) row; x segment and the activated 2x-rate
// signal live in LDS, so HBM sees one read and one write per element.
constexpr int A1_T = 1024;   // outputs per workgroup: 4 consecutive ones per thread
template <bool F32>
__global__ __launch_bounds__(256) void act1d_kernel(const void* __restrict__ x, void* __restrict__ y,
                                                    const float* __restrict__ filt, const float* __restrict__ alpha,
                                                    const float* __restrict__ beta, int logscale, int C, int L) {
    // xs[j] = x_pad[i0 + j] (replicate padding folded in);  as[j] = act(u[clamp(2 t0 - 5 + j)])
    __shared__ __attribute__((aligned(16))) float xs[A1_T + 16];
    __shared__ __attribute__((aligned(16))) float as[2 * A1_T + 16];
    const int row = blockIdx.x;                     // b * C + c (rows on grid x: B * C may exceed grid y's 65535)
    const int c = row % C;
    const int t0 = blockIdx.y * A1_T;
    const int64_t base = (int64_t)row * L;
    float f[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) f[j] = filt[j];    // uniform: scalar loads
    // u index n = 2 t0 - 5 + j uses x_pad[(n + 5) >> 1 .. + 5]; j = 0 -> x_pad index t0: take i0 = t0 - 1 as slot 0
    const int i0 = t0 - 1;
    for (int j = threadIdx.x; j < A1_T + 16; j += 256) {
        const int xi = min(max(i0 + j - 5, 0), L - 1);   // x_pad[i] = x[clamp(i - 5)]
        xs[j] = ld1<F32>(x, base + xi);
    }
    __syncthreads();
    // alpha == NULL: ELU in place of the snake (Oobleck units with antialias_activation and use_snake=False, autoencoders.py:24-37)
    const bool elu = alpha == nullptr;
    float a = elu ? 0.f : alpha[c], b = elu ? 1.f : beta[c];
    if (logscale && !elu) { a = __expf(a); b = __expf(b); }
    const float inv_b = 1.f / (b + 1e-9f);
    // each thread produces 8 consecutive as[] slots (+ the 12-slot tail by the first threads) from a register window of x
    for (int j0 = 8 * threadIdx.x; j0 < 2 * A1_T + 12; j0 += 8 * 256) {
        // slots j0..j0+7 <-> n = 2 t0 - 5 + j0 + e; x_pad index of tap q: ((n + 5) >> 1) + q = t0 + ((j0 + e) >> 1) + q
        float xw[12];
        const int w0 = (j0 >> 1) + 1;               // slot of x_pad[t0 + (j0 >> 1)] in xs (i0 = t0 - 1)
#pragma unroll
        for (int q = 0; q < 10; ++q) xw[q] = xs[min(w0 + q, A1_T + 15)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = j0 + e;
            if (j >= 2 * A1_T + 12) break;
            const int n = 2 * t0 - 5 + j;
            // tap of x_pad[t0 + (j >> 1) + q] is n + 15 - 2 i = 10 + (j & 1) - 2 q: even slots use taps 10, 8, .., 0 and odd
            // slots 11, 9, .., 1 (e and j have the same parity: j0 is a multiple of 8)
            float u = 0.f;
#pragma unroll
            for (int q = 0; q < 6; ++q) u += xw[(e >> 1) + q] * f[10 + (e & 1) - 2 * q];
            u *= 2.f;
            // replicate padding of the up-sampled signal: slots whose n falls outside [0, 2L) copy the edge value
            const int nc = min(max(n, 0), 2 * L - 1);
            if (nc != n) {
                const int ilo = (nc + 5) >> 1;      // recompute for the clamped index (rare: only at the two ends)
                float uu = 0.f;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int i = ilo + q, tap = nc + 15 - 2 * i;
                    const int sl = min(max(i - i0, 0), A1_T + 15);
                    uu += xs[sl] * f[tap];
                }
                u = 2.f * uu;
            }
            const float sn = fast_sin(u * a);
            as[j] = elu ? (u > 0.f ? u : __expf(u) - 1.f) : u + inv_b * sn * sn;
        }
    }
    __syncthreads();
    // 4 consecutive outputs per thread: y[t] = sum_j f[j] as[2 (t - t0) + j]  -> 18 consecutive slots, 16-B aligned
    const int tl = 4 * threadIdx.x;
    float w[20];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&as[2 * tl + 4 * q]);
        w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int t = t0 + tl + e;
        if (t >= L) break;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 12; ++j) acc += f[j] * w[2 * e + j];
        st1<F32>(y, base + t, acc);
    }
}

}  // namespace

extern "C" int kalle_act1d_fwd(const void* x, void* y, int dtype, const float* filter12, const float* alpha,
                               const float* beta, int logscale, int B, int C, int L, void* stream) {
    if (!x || !y || !filter12 || ((alpha == nullptr) != (beta == nullptr)) || B <= 0 || C <= 0 || L <= 0 ||
        (int64_t)B * C > 0x7fffffff || (L + A1_T - 1) / A1_T > 65535)
        return KALLE_ERR_ARG;
    dim3 grid(B * C, (L + A1_T - 1) / A1_T), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == KALLE_F32) KALLE_LAUNCH((act1d_kernel<true>), grid, block, 0, st, x, y, filter12, alpha, beta, logscale, C, L);
    else KALLE_LAUNCH((act1d_kernel<false>), grid, block, 0, st, x, y, filter12, alpha, beta, logscale, C, L);
    return kalle_check_launch();
}

extern "C" int kalle_weight_norm_fold(const float* v, const float* g, float* w_packed, int d0, int d1, int ksize,
                                      int transposed, void* stream) {
    if (!v || !w_packed || d0 <= 0 || d1 <= 0 || ksize <= 0) return KALLE_ERR_ARG;
    KALLE_LAUNCH(wn_fold_kernel, dim3((transposed & 1) ? d0 : ((d0 + 7) & ~7)), dim3(256), 0, static_cast<hipStream_t>(stream), v, g, w_packed, d0, d1,
                       ksize, transposed);
    return kalle_check_launch();
}


namespace {
struct ActArgs { int code = 0; const float* alpha = nullptr; const float* beta = nullptr; int logscale = 0; float param = 0.f; };
bool read_act(const kalle_act* a, ActArgs& o, bool input_side) {
    if (!a) return true;
    o.code = a->code; o.alpha = a->alpha; o.beta = a->beta; o.logscale = a->logscale; o.param = a->param;
    if (o.code < 0 || o.code > (input_side ? 4 : 3)) return false;
    if (o.code == 1 && (!o.alpha || !o.beta)) return false;
    return true;
}
bool fill_params(ConvParams& p, const kalle_act* in_act, const kalle_conv_epilogue* epi) {
    ActArgs ia, pa;
    if (!read_act(in_act, ia, true)) return false;
    p.act = ia.code; p.aa = ia.alpha; p.ab = ia.beta; p.logscale = ia.logscale; p.act_param = ia.param;
    p.res = nullptr; p.out_scale = 1.f; p.post = 0; p.y_raw = nullptr;
    if (epi) {
        if (!read_act(&epi->post_act, pa, false)) return false;
        p.res = epi->residual;
        p.y_raw = epi->y_raw;
        p.out_scale = epi->out_scale;
        p.post = (epi->tanh ? 1 : 0) | (epi->accumulate ? 2 : 0);
    }
    p.pact = pa.code; p.paa = pa.alpha; p.pab = pa.beta; p.plogscale = pa.logscale; p.pparam = pa.param;
    return true;
}
// v2 tile choice: (channels per wave, positions per lane).  A workgroup's cost is its padded output count times a
// penalty for the shorter (less well pipelined) tiles; workgroups beyond one per CU run in further rounds.
struct TileChoice { int cow, lpt; };
TileChoice pick_tile(int64_t npos, int Cout, int B, int halo, int span, int lpt_only, int64_t wg_mult) {
    const int cows[4] = {16, 8, 8, 8}, lpts[4] = {8, 8, 4, 2};
    const double pen[4] = {1.0, 1.08, 1.2, 1.5};
    TileChoice best{0, 0};
    double best_cost = 0;
    for (int i = 0; i < 4; ++i) {
        const int lt = 64 * lpts[i], cot = 4 * cows[i];
        if (lt + halo > span || (lpt_only && lpts[i] != lpt_only)) continue;
        if (cows[i] == 16 && (Cout < 64 || ((Cout + 7) & ~7) % 16)) continue;   // (16-wide weight reads need CoutP % 16 == 0)
        const int64_t nwg = ((npos + lt - 1) / lt) * ((Cout + cot - 1) / cot) * B * wg_mult;
        // rounds: 2 (COW 16) / 3 (COW 8) workgroups fit a CU; a lone workgroup per CU still costs ~0.6 of a full round
        const int64_t slots = 256 * (cows[i] == 16 ? 2 : 3);
        const double rounds = nwg <= 256 ? 0.6 : (double)((nwg + slots - 1) / slots);
        const double cost = (double)lt * cot * pen[i] * rounds * (cows[i] == 16 ? 2 : 3);
        if (!best.cow || cost < best_cost) { best = TileChoice{cows[i], lpts[i]}; best_cost = cost; }
    }
    return best;
}
constexpr int V2_SPAN = 640;
}  // namespace

extern "C" int kalle_conv1d_fwd(const void* x, int x_dtype, const float* w_packed, const float* bias, void* y,
                                int y_dtype, int B, int Cin, int Lin, int Cout, int Lout, int ksize, int stride,
                                int padding, int dilation, const kalle_act* in_act, const kalle_conv_epilogue* epi,
                                void* stream) {
    if (!x || !w_packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Lin <= 0 || Lout <= 0) return KALLE_ERR_ARG;
    if (ksize <= 0 || ksize > MAX_K || stride <= 0 || dilation <= 0 || padding < 0) return KALLE_ERR_ARG;
    // `padding` is the LEFT pad; the right pad is implied by Lout (symmetric, 'same' or causal alike): taps beyond Lin read 0
    if ((int64_t)(Lout - 1) * stride - padding >= Lin) return KALLE_ERR_ARG;
    if (B > 65535 || (Cout + 7) / 8 > 65535) return KALLE_ERR_ARG;
    ConvParams p{};
    if (!fill_params(p, in_act, epi)) return KALLE_ERR_ARG;
    p.x = x; p.w = w_packed; p.bias = bias; p.y = y;
    p.B = B; p.Cin = Cin; p.Lin = Lin; p.Cout = Cout; p.Lout = Lout; p.K = ksize; p.stride = stride; p.pad = padding;
    p.dil = dilation; p.CoutP = (Cout + 7) & ~7; p.xC = p.act == 4 ? 2 * Cin : Cin;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool xf = x_dtype == KALLE_F32, yf = y_dtype == KALLE_F32;
    const int halo = (ksize - 1) * dilation;
    // weights small enough to stay in one XCD's L2: run the channel tiles of a position tile back to back (x re-read hits L2)
    p.co_fast = (int64_t)Cin * ksize * p.CoutP * 4 <= (2 << 20);
    const bool v2_stride = stride == 1 || (dilation == 1 && (stride == 2 || stride == 4 || stride == 8) && ksize <= 32);
    static const bool v1_env = getenv("KALLE_CONV_V1") != nullptr;       // experiment switch, read once per process
    if (v2_stride && p.act != 4 && xf == yf && !v1_env) {
#define KALLE_CONV_V2N(COW, LPT, WCO, CI, SPAN, NW)                                                                    \
    do {                                                                                                                \
        p.ntile = (Lout + 64 * LPT * (NW / WCO) - 1) / (64 * LPT * (NW / WCO));                                        \
        p.nco = (Cout + COW * WCO - 1) / (COW * WCO);                                                                   \
        if ((int64_t)p.ntile * p.nco > 0x7fffffff) return KALLE_ERR_ARG;                                                \
        dim3 g(p.ntile * p.nco, 1, B);                                                                                  \
        if (xf)                                                                                                         \
            KALLE_LAUNCH((conv1d_v2_kernel<COW, LPT, WCO, CI, SPAN, true, true, NW>), g, dim3(64 * NW), 0, st, p);      \
        else                                                                                                            \
            KALLE_LAUNCH((conv1d_v2_kernel<COW, LPT, WCO, CI, SPAN, false, false, NW>), g, dim3(64 * NW), 0, st, p);    \
        return kalle_check_launch();                                                                                    \
    } while (0)
#define KALLE_CONV_V2(COW, LPT, WCO, CI, SPAN) KALLE_CONV_V2N(COW, LPT, WCO, CI, SPAN, 4)
        if (stride == 1) {
            const TileChoice tc = pick_tile(Lout, Cout, B, halo, V2_SPAN, 0, 1);
            if (Cout <= 4) {
                if (halo + 512 <= V2_SPAN) KALLE_CONV_V2(2, 2, 1, 8, 640);
            } else if (tc.cow == 16) {
                // 8 waves share one staged x tile for 128 output channels (pointwise convs: 32-channel chunks to cover
                // the HBM latency; measured +15 % at C >= 256, +8 % on the k = 7 convs at C = 256, neutral at C = 128)
                // pointwise convs: 256 positions x 128 channels per workgroup (64 accumulator registers per wave, 66 KiB of LDS) so
                // that TWO workgroups share a CU - one's residual loads / stores run under the other's FMAs; with 512 positions
                // (128 accumulator registers, 133 KiB) a CU runs one workgroup whose memory phases nothing overlaps
                // (KALLE_CONV_K1_WIDE=1: the 512-position tile of rounds 1-2)
                static const bool k1_wide = getenv("KALLE_CONV_K1_WIDE") && atoi(getenv("KALLE_CONV_K1_WIDE")) == 1;
                if (ksize == 1 && Cout > 64 && !k1_wide) KALLE_CONV_V2N(16, 4, 8, 32, 256, 8);
                if (ksize == 1 && Cout > 64) KALLE_CONV_V2N(16, 8, 8, 32, 512, 8);
                // (the same halving for the wide k = 7 convs - 7 x the FMAs per byte - is worth 0.4 % of a decode: not taken)
                if (ksize != 1 && Cout >= 256 && halo + 512 <= 640) KALLE_CONV_V2N(16, 8, 8, 8, 640, 8);
                if (ksize == 1) KALLE_CONV_V2(16, 8, 4, 16, 512);   // pointwise conv: longer chunks cover the HBM latency
                KALLE_CONV_V2(16, 8, 4, 8, 640);
            } else {
                switch (tc.lpt) {
                    case 8: KALLE_CONV_V2(8, 8, 4, 8, 640);
                    case 4: KALLE_CONV_V2(8, 4, 4, 8, 640);
                    case 2: KALLE_CONV_V2(8, 2, 4, 8, 640);
                    default: break;
                }
            }
        } else if (Cout > 4) {                           // stride 2 / 4 / 8: de-interleaved staging, 1024/stride positions
            const bool wide = Cout >= 64 && p.CoutP % 16 == 0;
            if (stride == 2) KALLE_CONV_V2(8, 8, 4, 8, 1088);
            if (stride == 4) { if (wide) KALLE_CONV_V2(16, 4, 4, 8, 1088); else KALLE_CONV_V2(8, 4, 4, 8, 1088); }
            if (stride == 8 && (int64_t)Lout * B <= 1024) KALLE_CONV_V2(8, 2, 4, 8, 1088);   // longer: fallback is faster
        }
#undef KALLE_CONV_V2
#undef KALLE_CONV_V2N
    }
    if ((L_T - 1) * stride + halo + 1 > MAX_SPAN) return KALLE_ERR_UNSUPPORTED;
    dim3 grid((Lout + L_T - 1) / L_T, (Cout + CO_T - 1) / CO_T, B), block(256);
    if (xf && yf) KALLE_LAUNCH((conv1d_kernel<true, true>), grid, block, 0, st, p);
    else if (xf) KALLE_LAUNCH((conv1d_kernel<true, false>), grid, block, 0, st, p);
    else if (yf) KALLE_LAUNCH((conv1d_kernel<false, true>), grid, block, 0, st, p);
    else KALLE_LAUNCH((conv1d_kernel<false, false>), grid, block, 0, st, p);
    return kalle_check_launch();
}

extern "C" int kalle_conv_pad_len(int Lout, int ksize, int stride, int padding, int dilation) {
    if (stride == 1) return ((Lout + 15) & ~15) + (ksize - 1) * dilation;
    const int padq = (padding + stride - 1) / stride, d = padq * stride - padding;   // phase rows of Lq slots each
    return stride * (((Lout + 15) & ~15) + (ksize - 1 + d) / stride + 1);
}

extern "C" int kalle_conv_pad_act(const float* x, float* x_padded, int B, int C, int Lin, int Lp, int padding,
                                  const kalle_act* act, int phases, void* stream) {
    if (!x || !x_padded || B <= 0 || C <= 0 || Lin <= 0 || Lp <= 0 || padding < 0 || phases <= 0 || Lp % phases)
        return KALLE_ERR_ARG;
    const int gx = (Lp + 255) / 256 > 64 ? 64 : (Lp + 255) / 256;
    if ((int64_t)B * C * gx > 0x7fffffff) return KALLE_ERR_ARG;
    ActArgs a;
    if (!read_act(act, a, false)) return KALLE_ERR_ARG;
    KALLE_LAUNCH(pad_act_kernel, dim3((unsigned)(B * C * gx)), dim3(256), 0,
                 static_cast<hipStream_t>(stream), x, x_padded, C, Lin, Lp, padding, a.code, a.alpha, a.beta, a.logscale, a.param,
                 phases, gx);
    return kalle_check_launch();
}

// workgroup-level input-channel split of the channels-per-lane conv: with few positions AND few output channels even 4
// waves per tile leave most SIMDs idle while every wave walks Cin * K dependent prefetch steps (the 1024 -> 2048 stride-8
// and 2048 -> 128 convs at the bottom of a single-clip encode: 112 / 14 position-channel tiles)
static int cfirst_ksplit(int B, int Cin, int Cout, int Lout, int ksize) {
    const int64_t waves = (int64_t)((Lout + 15) / 16) * ((Cout + 255) / 256) * B * 4;
    int ks = 1;
    while (ks < 16 && waves * ks < 4096 && (int64_t)Cin * ksize / (8 * ks) >= 64) ks *= 2;
    return ks;
}
extern "C" int kalle_conv_cfirst_ws_floats(int B, int Cin, int Cout, int Lout, int ksize) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || ksize <= 0) return KALLE_ERR_ARG;
    const int ks = cfirst_ksplit(B, Cin, Cout, Lout, ksize);
    const int64_t n = ks > 1 ? (int64_t)ks * B * Cout * Lout : 0;
    return n > 0x7fffffff ? 0 : (int)n;
}

extern "C" int kalle_conv1d_cfirst_fwd(const float* x_padded, const float* w_packed, const float* bias, float* y, int B,
                                       int Cin, int Lp, int Cout, int Lout, int ksize, int stride, int padding,
                                       int dilation, const kalle_conv_epilogue* epi, float* workspace, void* stream) {
    if (!x_padded || !w_packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || ksize <= 0 || dilation <= 0 ||
        stride <= 0 || padding < 0)
        return KALLE_ERR_ARG;
    if (stride > 1 && dilation != 1) return KALLE_ERR_UNSUPPORTED;
    if (Lp < kalle_conv_pad_len(Lout, ksize, stride, padding, dilation) || (stride > 1 && Lp % stride) || B > 65535)
        return KALLE_ERR_ARG;
    ConvParams q{};
    if (!fill_params(q, nullptr, epi)) return KALLE_ERR_ARG;
    // waves without splitting the input channels; aim for >= 2048 (two per SIMD)
    const int64_t waves = (int64_t)((Lout + 15) / 16) * ((Cout + 255) / 256) * B;
    const int split = waves >= 2048 ? 1 : (waves >= 1024 ? 2 : 4);
    const int ks = workspace ? cfirst_ksplit(B, Cin, Cout, Lout, ksize) : 1;   // (the caller sized it with ..._ws_floats)
    // stride > 1: x_padded is de-interleaved into `stride` phase rows (left pad padq * stride): tap k of output l reads
    // phase (k + d) % stride at slot l + (k + d) / stride
    const int padq = (padding + stride - 1) / stride, d = padq * stride - padding;
    const int Lq = stride > 1 ? Lp / stride : 0;
    ConvCParams p{x_padded, w_packed, bias, static_cast<const float*>(q.res), y, static_cast<float*>(q.y_raw), B, Cin, Lp, Cout, (Cout + 7) & ~7, Lout, ksize,
                  dilation, q.post, split, ks, workspace, 1, Lout, dilation, 1, 0, stride > 1 ? d * Lq : 0, stride, Lq,
                  stride > 1 ? d : 0, q.out_scale, q.pact, q.paa, q.pab, q.plogscale, q.pparam};
    if (p.CoutP < 4) return KALLE_ERR_UNSUPPORTED;
    const int tiles_per_wg = 4 / split;
    if ((int64_t)B * ks > 65535) return KALLE_ERR_ARG;
    dim3 grid(((Lout + 15) / 16 + tiles_per_wg - 1) / tiles_per_wg, (Cout + 255) / 256, B * ks);
    KALLE_LAUNCH(conv1d_cfirst_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    if (ks > 1) {
        const int64_t n = (int64_t)B * Cout * Lout;
        KALLE_LAUNCH(cfirst_finish_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
    }
    return kalle_check_launch();
}

extern "C" int kalle_convT_pad_len(int Lout, int ksize, int stride, int padding) {
    const int nq = (Lout - 1 + padding) / stride + 1, mmax = (ksize + stride - 1) / stride;
    return ((nq + 15) & ~15) + mmax - 1;
}

extern "C" int kalle_conv_transpose1d_cfirst_fwd(const float* x_padded, const float* w_packed, const float* bias, float* y,
                                                 int B, int Cin, int Lp, int Cout, int Lout, int ksize, int stride,
                                                 int padding, const kalle_conv_epilogue* epi, void* stream) {
    if (!x_padded || !w_packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Lout <= 0 || ksize <= 0 || stride <= 0 ||
        padding < 0)
        return KALLE_ERR_ARG;
    if (Lp < kalle_convT_pad_len(Lout, ksize, stride, padding) || B > 65535) return KALLE_ERR_ARG;
    ConvParams q{};
    if (!fill_params(q, nullptr, epi)) return KALLE_ERR_ARG;
    const int nq = (Lout - 1 + padding) / stride + 1, mmax = (ksize + stride - 1) / stride;
    const int64_t waves = (int64_t)((nq + 15) / 16) * ((Cout + 255) / 256) * B * stride;
    const int split = waves >= 2048 ? 1 : (waves >= 1024 ? 2 : 4);
    // x_padded has mmax-1 leading zeros: tap m of input position q reads slot q + (mmax-1) - m
    ConvCParams p{x_padded, w_packed, bias, static_cast<const float*>(q.res), y, static_cast<float*>(q.y_raw), B, Cin, Lp, Cout, (Cout + 7) & ~7, Lout, ksize,
                  1, q.post, split, 1, nullptr, stride, nq, -1, stride, padding, mmax - 1, 1, 0, 0, q.out_scale, q.pact, q.paa,
                  q.pab, q.plogscale, q.pparam};
    if (p.CoutP < 4) return KALLE_ERR_UNSUPPORTED;
    const int tiles_per_wg = 4 / split;
    const int64_t gx = (int64_t)(((nq + 15) / 16 + tiles_per_wg - 1) / tiles_per_wg) * stride;
    if (gx > 0x7fffffff) return KALLE_ERR_ARG;
    dim3 grid((unsigned)gx, (Cout + 255) / 256, B);
    KALLE_LAUNCH(conv1d_cfirst_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return kalle_check_launch();
}

extern "C" int kalle_conv_transpose1d_fwd(const void* x, int x_dtype, const float* w_packed, const float* bias, void* y,
                                          int y_dtype, int B, int Cin, int Lin, int Cout, int Lout, int ksize,
                                          int stride, int padding, const kalle_act* in_act,
                                          const kalle_conv_epilogue* epi, void* stream) {
    if (!x || !w_packed || !y || B <= 0 || Cin <= 0 || Cout <= 0 || Lin <= 0 || Lout <= 0) return KALLE_ERR_ARG;
    if (ksize <= 0 || stride <= 0 || padding < 0) return KALLE_ERR_ARG;
    // shorter than (Lin-1)*stride - 2*padding + K = causal trim of the tail; up to `padding` longer = the outputs the symmetric
    // trim would drop on the right (the data gradient of a strided conv whose input length is not a multiple of the stride)
    if (Lout > (Lin - 1) * stride - padding + ksize) return KALLE_ERR_ARG;
    if (B > 65535 || (Cout + 7) / 8 > 65535) return KALLE_ERR_ARG;
    ConvParams p{};
    if (!fill_params(p, in_act, epi)) return KALLE_ERR_ARG;
    if (p.act == 4) return KALLE_ERR_UNSUPPORTED;
    p.x = x; p.w = w_packed; p.bias = bias; p.y = y;
    p.B = B; p.Cin = Cin; p.Lin = Lin; p.Cout = Cout; p.Lout = Lout; p.K = ksize; p.stride = stride; p.pad = padding;
    p.dil = 1; p.CoutP = (Cout + 7) & ~7; p.xC = Cin;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool xf = x_dtype == KALLE_F32, yf = y_dtype == KALLE_F32;
    const int mmax = (ksize + stride - 1) / stride;
    static const bool v1_env = getenv("KALLE_CONV_V1") != nullptr;
    if (xf == yf && mmax <= 64 && !v1_env) {
        const int nq = (Lout - 1 + padding) / stride + 1;      // input positions that reach an output
#define KALLE_CONVT_V2(COW, LPT, WCO)                                                                                  \
    do {                                                                                                                \
        p.ntile = (nq + 64 * LPT * (4 / WCO) - 1) / (64 * LPT * (4 / WCO));                                             \
        p.nco = (Cout + COW * WCO - 1) / (COW * WCO);                                                                   \
        const int64_t nwg = (((int64_t)p.ntile * p.nco + 7) / 8) * 8 * stride;                                          \
        if (nwg > 0x7fffffff) return KALLE_ERR_ARG;                                                                     \
        dim3 g((unsigned)nwg, 1, B);                                                                                    \
        if (xf) KALLE_LAUNCH((convT1d_v2_kernel<COW, LPT, WCO, 8, 640, true, true>), g, dim3(256), 0, st, p);           \
        else KALLE_LAUNCH((convT1d_v2_kernel<COW, LPT, WCO, 8, 640, false, false>), g, dim3(256), 0, st, p);            \
        return kalle_check_launch();                                                                                    \
    } while (0)
        const TileChoice tc = pick_tile(nq, Cout, B, mmax - 1, V2_SPAN, 0, stride);
        if (Cout <= 4) {
            if (mmax - 1 + 512 <= V2_SPAN) KALLE_CONVT_V2(2, 2, 1);
        } else if (tc.cow == 16) {
            KALLE_CONVT_V2(16, 8, 4);
        } else {
            switch (tc.lpt) {
                case 8: KALLE_CONVT_V2(8, 8, 4);
                case 4: KALLE_CONVT_V2(8, 4, 4);
                case 2: KALLE_CONVT_V2(8, 2, 4);
                default: break;
            }
        }
#undef KALLE_CONVT_V2
    }
    if (ksize > MAX_K + 2 || ksize > 2 * stride + 1 || p.res || p.post || p.pact || p.y_raw || p.out_scale != 1.f)
        return KALLE_ERR_UNSUPPORTED;
    dim3 grid((Lout + L_T - 1) / L_T, (Cout + CO_T - 1) / CO_T, B), block(256);
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
