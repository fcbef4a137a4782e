// gemm2: the large-shape bf16 MFMA GEMM (gfx950) + the C-ABI dispatcher kalle_gemm_bf16.
//
// 8 waves (512 threads) per workgroup, one workgroup per CU, tile (WM*TM*16) x (WN*64) x 64 with 64x64 or 128x64
// per wave; operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging); a ring of three LDS stages
// keeps the DMA of two K-tiles in flight across every barrier (counted vmcnt, never drained);
// fragments for the NEXT 32-deep k-step are read into a second register set while the MFMAs of the current k-step
// issue, so LDS latency, the once-per-K-tile barrier and the DMA of the tile after next are all covered by matrix
// work.  LDS images are lane-linear (an LDS-DMA wave-instruction writes 1 KiB contiguously), so the bank-conflict
// swizzles are applied to the per-lane SOURCE address and undone by the same involution on the fragment read:
//   k-contiguous operand [R][64]: 16-B chunk c of row r stored at slot c ^ ((r>>1)&7)            (ds_read_b128)
//   k-major operand [64][R]:      k-row k stored at position pos(k) (odd 8-blocks: rows 0-3 <-> 4-7), chunk c of
//                                 position p at slot c ^ (2*(p&7))                                (ds_read_b64_tr_b16)
// Split-K (wgrad: the output is weight-shaped, the contraction runs over all tokens) turns the grid into
// tiles x splits with fp32 atomics into a zeroed C - this is what fills 256 CUs when the output has < 256 tiles.
// Requirements: K % 8 == 0 (a ragged last K-tile takes its missing 16-byte pieces from a block of zeros).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <type_traits>
#include <utility>

#include "gemm_common.h"
#include "../../include/kalle_hip.h"

namespace {

constexpr int BK2 = 64;
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 64 zero bytes: the source of every 16-byte piece of a ragged last K-tile that lies beyond K (an LDS-DMA load has no range
// check and no per-lane predicate, but its SOURCE address is per lane)
__device__ __attribute__((aligned(64))) const unsigned kalle_zero_block[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// ---- LDS-DMA issue for one operand region -------------------------------------------------------------------------
// R: tile extent (rows of a k-contiguous operand / contiguous extent of a k-major one); NW waves share R/8 1-KiB pieces
template <bool KM, int R, int NW>
struct Loader {
    static constexpr int PER_WAVE = (R / 8) / NW;   // 1-KiB wave-instructions per wave per K-tile
    const bf16_t* src[PER_WAVE];                    // per-lane source pointer of each piece at the current K-tile
    int64_t kstep;                                  // elements to advance per K-tile

    // glu_inner > 0 (k-contiguous B of the fused SwiGLU forward): the 256 tile rows are, per 64-row wave slice,
    // 32 rows of the x half followed by the 32 matching rows of the gate half of the [2*inner][K] GLU weight
    __device__ __forceinline__ void init(const bf16_t* X, int64_t ld, int Rtot, int r0, int k0, int wave, int lane,
                                         int glu_inner = 0, int glu_tn = 0) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave * PER_WAVE + i;
            if constexpr (!KM) {
                const int row = 8 * q + (lane >> 3);
                const int slot = lane & 7;
                const int c = slot ^ ((row >> 1) & 7);
                int gr = min(r0 + row, Rtot - 1);                           // clamp: rows >= Rtot are never stored
                if (glu_inner > 0) {
                    const int wn_ = row >> 6, w_ = row & 63;
                    gr = (w_ >= 32 ? glu_inner : 0) + glu_tn * (R / 2) + wn_ * 32 + (w_ & 31);
                }
                src[i] = X + (int64_t)gr * ld + k0 + 8 * c;
            } else {
                constexpr int RB = 2 * R;                                   // bytes per k-row
                constexpr int ROWS = 1024 / RB;                             // k-rows per piece
                const int pos = q * ROWS + (lane * 16) / RB;
                const int slot = ((lane * 16) % RB) / 16;
                const int c = slot ^ (2 * (pos & 7));
                const int krow = (pos & ~7) | ((pos & 7) ^ (((pos >> 3) & 1) << 2));
                const int gc = min(r0 + 8 * c, Rtot - 8);
                src[i] = X + (int64_t)(k0 + krow) * ld + gc;
            }
        }
        kstep = KM ? 64 * ld : 64;
    }
    __device__ __forceinline__ void issue(char* region, int wave) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            __builtin_amdgcn_global_load_lds(GPTR(src[i]), LDS_PTR(void, region + (wave * PER_WAVE + i) * 1024), 16, 0, 0);
            src[i] += kstep;
        }
    }
    // one piece (the software-pipelined 256 x 256 loop spreads a K-tile's pieces over its MFMA rows); `tail`: this is the ragged
    // last K-tile (see issue_tail)
    template <int I>
    __device__ __forceinline__ void issue_one(char* region, int wave, int lane, bool tail, int kvalid) {
        static_assert(I < PER_WAVE, "piece index");
        const bf16_t* sp = src[I];
        if (tail) {
            const int q = wave * PER_WAVE + I;
            bool ok;
            if constexpr (!KM) {
                const int row = 8 * q + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                ok = 8 * c < kvalid;
            } else {
                constexpr int RB = 2 * R, ROWS = 1024 / RB;
                const int pos = q * ROWS + (lane * 16) / RB;
                const int krow = (pos & ~7) | ((pos & 7) ^ (((pos >> 3) & 1) << 2));
                ok = krow < kvalid;
            }
            if (!ok) sp = reinterpret_cast<const bf16_t*>(kalle_zero_block);
        }
        __builtin_amdgcn_global_load_lds(GPTR(sp), LDS_PTR(void, region + (wave * PER_WAVE + I) * 1024), 16, 0, 0);
        src[I] += kstep;
    }
    // the ragged last K-tile (K % 64 != 0): only the first `kvalid` (a multiple of 8) of its 64 k exist; every piece that lies
    // beyond them is fetched from the zero block instead, so both operands contribute exact zeros there
    __device__ __forceinline__ void issue_tail(char* region, int wave, int lane, int kvalid) {
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int q = wave * PER_WAVE + i;
            bool ok;
            if constexpr (!KM) {
                const int row = 8 * q + (lane >> 3);
                const int c = (lane & 7) ^ ((row >> 1) & 7);
                ok = 8 * c < kvalid;
            } else {
                constexpr int RB = 2 * R, ROWS = 1024 / RB;
                const int pos = q * ROWS + (lane * 16) / RB;
                const int krow = (pos & ~7) | ((pos & 7) ^ (((pos >> 3) & 1) << 2));
                ok = krow < kvalid;
            }
            const bf16_t* sp = ok ? src[i] : reinterpret_cast<const bf16_t*>(kalle_zero_block);
            __builtin_amdgcn_global_load_lds(GPTR(sp), LDS_PTR(void, region + (wave * PER_WAVE + i) * 1024), 16, 0, 0);
            src[i] += kstep;
        }
    }
    // tile index `t` of this workgroup's K range; `tail_t` = index of the ragged tile (-1: none)
    __device__ __forceinline__ void issue_at(char* region, int wave, int lane, int t, int tail_t, int kvalid) {
        if (t == tail_t) issue_tail(region, wave, lane, kvalid);
        else issue(region, wave);
    }
};

// ---- fragment reads: inline asm, counted by hand ----------------------------------------------------------------
// hipcc (ROCm 7.2) drains the LDS-DMA queue (s_waitcnt vmcnt(0)) in front of every ds_read_tr builtin and, when the
// loop spans basic blocks, waits lgkmcnt(0) for reads issued a moment ago - both serialise the pipeline.  The reads
// are therefore issued from asm statements the compiler does not count; the explicit waits below do the counting.
template <int OFF>
__device__ __forceinline__ i32x4 lds_b128(unsigned a) {
    i32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "n"(OFF));
    return r;
}
__device__ __forceinline__ i32x2 lds_tr(unsigned a) {
    i32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(a));
    return r;
}

// per-lane LDS byte offsets of an operand's fragments inside one pipeline stage
template <bool KM, int R>
struct Reader {
    unsigned b0[2], b1[2];   // [k-step]; k-contiguous: b0 only. k-major: the two transposed reads of a fragment
    __device__ __forceinline__ void init(int region_off, int rbase, int lane) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if constexpr (!KM) {
                const int row = rbase + (lane & 15);
                const int c = 4 * s + (lane >> 4);
                b0[s] = region_off + row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
                b1[s] = 0;
            } else {
                constexpr int RB = 2 * R;
                const int g = lane >> 4, i = lane & 15;
                const int q = i >> 2, pp = i & 3;
                const int b = 4 * s + g;
                const int odd = b & 1;
                const int pos1 = 8 * b + (odd ? 4 : 0) + q;
                const int pos2 = 8 * b + (odd ? 0 : 4) + q;
                const int cl = (rbase >> 3) + (pp >> 1);          // rbase % 64 == 0: tile t adds 2t to bits 1-2
                const int hb = (pp & 1) * 8;
                b0[s] = region_off + pos1 * RB + ((cl ^ (2 * (pos1 & 7))) << 4) + hb;
                b1[s] = region_off + pos2 * RB + ((cl ^ (2 * (pos2 & 7))) << 4) + hb;
            }
        }
    }
    static constexpr int NI = KM ? 2 : 1;    // LDS instructions per fragment
    // ONE fragment: tile T (16 operand rows from rbase + 16 T) of k-step S, stage byte offset `so`
    template <int S, int T>
    __device__ __forceinline__ void read1(unsigned so, i32x4& f) const {
        if constexpr (!KM) {
            f = lds_b128<2048 * T>(b0[S] + so);
        } else {
            const i32x2 lo = lds_tr((b0[S] + so) ^ (T << 5));
            const i32x2 hi = lds_tr((b1[S] + so) ^ (T << 5));
            f = i32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    }
    // fragments of the 4 tiles (16 operand rows each) of k-step S, stage byte offset `so`
    // NTILE tiles starting at tile T0 (16 operand rows each, from rbase)
    template <int S, int NTILE, int T0 = 0>
    __device__ __forceinline__ void read(unsigned so, i32x4 (&f)[NTILE]) const {
        if constexpr (!KM) {
            const unsigned a = b0[S] + so;
            f[0] = lds_b128<2048 * (T0 + 0)>(a);
            f[1] = lds_b128<2048 * (T0 + 1)>(a);
            f[2] = lds_b128<2048 * (T0 + 2)>(a);
            f[3] = lds_b128<2048 * (T0 + 3)>(a);
            if constexpr (NTILE == 8) {
                f[4] = lds_b128<2048 * (T0 + 4)>(a);
                f[5] = lds_b128<2048 * (T0 + 5)>(a);
                f[6] = lds_b128<2048 * (T0 + 6)>(a);
                f[7] = lds_b128<2048 * (T0 + 7)>(a);
            }
        } else {
            const unsigned a0 = b0[S] + so, a1 = b1[S] + so;
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                // (cl + 2t) ^ 2u == (cl ^ 2u) ^ 2t for cl % 8 == 0 -> the tile index is an XOR on address bits 5-7
                const i32x2 lo = lds_tr(a0 ^ ((T0 + t) << 5));
                const i32x2 hi = lds_tr(a1 ^ ((T0 + t) << 5));
                f[t] = i32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        }
    }
};

// diagnostics: wave 0 / wave 7 of a workgroup leave a time stamp (constant 100 MHz counter, comparable across CUs)
__device__ __forceinline__ void gemm_stamp(const GemmParams& p, int wave, int lane, int idx, int slot = -1) {
    if (p.stamps && (wave == 0 || wave == 7) && lane == 0)
        p.stamps[((int64_t)(slot < 0 ? (int)blockIdx.x : slot) * 2 + (wave ? 1 : 0)) * 8 + idx] = __builtin_amdgcn_s_memrealtime();
}

// ---- epilogue: per wave, one 16 x 64 fp32 piece at a time through a wave-private LDS patch ---------------------------
// What the in-kernel stamps of round 3 showed (tools/gemm_stamps.py, all 256 CUs in their epilogue at once): the STORE side is
// absorbed by L2 / Infinity Cache at fabric speed and is bound by the number of store instructions (8-byte bf16 stores: 128 KB
// in 8.8 us; 16-byte stores: 192 KB in 5.3 us), the LOAD side (fp32 residual, saved SwiGLU pre-activations: 256 KB per tile from
// HBM) by the bytes a wave keeps in flight - one 16-row piece ahead is 4 KB per wave = 16 GB/s per CU against ~2 us of loaded
// latency - or so it seemed: fetching THREE pieces ahead (first loads before the hand-over barrier) changed nothing (fused
// SwiGLU backward 20.5 -> 18.4 us, fp32 residual 20.0 -> 22.3 us per tile), because with every CU in its epilogue at once the
// 64 MB read + 64 MB written per round ARE the HBM's 6.5-7 TB/s for those 18-20 us.  So: every output leaves in 16-byte stores
// (bf16 outputs 8.8 -> 6.4 us per tile), and the epilogue operands stay one piece ahead (EPI_DEPTH; deeper only costs registers).
constexpr int EPI_DEPTH = 1;

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// patch: 16 rows x 64 floats per wave, NO padding (8 waves = exactly the 32 KiB the two 64-KiB stages leave of a CU's 160 KiB, so
// the persistent 256 x 256 kernel can refill the stages while the epilogue runs); 16-byte chunk c of row r sits at chunk
// c ^ (r & 1): conflict-free for the fp32 (rows g + 4i, chunk li) and the bf16 (row lane >> 3, chunks 2j, 2j + 1) read patterns
__device__ __forceinline__ int pidx(int r, int c) { return r * 64 + (c ^ ((r & 1) << 2)); }

template <bool C_F32, int TM, int GLU = 0, typename Hook = NoHook>
__device__ __forceinline__ void wave_epilogue(const GemmParams& p, f32x4 (&acc)[TM][4], char* patch_base, int wave, int lane,
                                              int row0, int col0, Hook after_barrier = Hook{}, int stamp_slot = -1) {
    constexpr int DEPTH = EPI_DEPTH < TM ? EPI_DEPTH : TM;
    float* patch = reinterpret_cast<float*>(patch_base) + wave * (16 * 64);
    const int g = lane >> 4, li = lane & 15;
    // The residual (the fp32 stream, updated in place: it aliases C as far as the compiler can tell) is fetched DEPTH 16-row
    // pieces AHEAD of the stores: left inside the store loop every one of a wave's 32 load -> add -> store chains exposes a
    // global-load latency (the pieces are disjoint rows, so reading a later piece before an earlier one is stored is safe).
    const bool pre_res = C_F32 && GLU == 0 && !p.atomic && p.residual != nullptr;
    f32x4 rbuf[DEPTH][4];
    auto load_res = [&](int mt, f32x4 (&rv)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = row0 + 16 * mt + g + 4 * i, gn = col0 + 4 * li;
            rv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (gm < p.M && gn < p.N) rv[i] = *reinterpret_cast<const f32x4*>(p.residual + gemm_crow(p, gm) * p.ldr + gn);
        }
    };
    // fused SwiGLU backward: the saved pre-activations h (x | gate) of a piece, fetched ahead like the residual
    i32x4 hbuf[DEPTH][2][2];
    auto load_h = [&](int mt, i32x4 (&hv)[2][2]) {
        const int c8 = (lane & 7) * 8, j0 = col0 + c8;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int gm = row0 + 16 * mt + (lane >> 3) + 8 * half;
            hv[half][0] = i32x4{0, 0, 0, 0};
            hv[half][1] = i32x4{0, 0, 0, 0};
            if (gm < p.M && j0 < p.glu_inner && !(p.dbg & 1)) {
                const bf16_t* hp = static_cast<const bf16_t*>(p.glu_aux) + (int64_t)gm * 2 * p.glu_inner;
                hv[half][0] = *reinterpret_cast<const i32x4*>(hp + j0);
                hv[half][1] = *reinterpret_cast<const i32x4*>(hp + p.glu_inner + j0);
            }
        }
    };
    // (the LDS-DMA of the main loop was waited for at its last hand-over: nothing of the pipeline is outstanding here - only
    // the fragment reads must have returned)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // all waves are done with the pipeline stages
    after_barrier();                // (persistent kernel: the stage of the last K-tile is free now - DMA of the next tile's K-tile 1)
    gemm_stamp(p, wave, lane, 3, stamp_slot);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if constexpr (GLU == 2) load_h(d, hbuf[d]);
        if (pre_res) load_res(d, rbuf[d]);
    }
    float sx[8], sg[8];             // GLU backward: bias-gradient partial sums of this lane's 8 columns
#pragma unroll
    for (int e = 0; e < 8; ++e) { sx[e] = 0.f; sg[e] = 0.f; }
    // the bias of a lane's columns is the same for every piece: loaded once (inside the loop each load sits behind the
    // previous piece's stores, which may alias it for all the compiler knows)
    f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f}, bias8[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float bx[8], bg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bx[e] = 0.f; bg[e] = 0.f; }
    if (p.bias) {
        if constexpr (GLU == 1) {
            const int j0 = col0 + (lane & 3) * 8;
            if (j0 < p.glu_inner) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { bx[e] = p.bias[j0 + e]; bg[e] = p.bias[p.glu_inner + j0 + e]; }
            }
        } else if constexpr (GLU == 0) {
            if constexpr (C_F32) {
                if (!p.atomic && col0 + 4 * li < p.N) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col0 + 4 * li);
            } else {
                const int gn = col0 + 8 * (lane & 7);
                if (gn < p.N) {
                    bias8[0] = *reinterpret_cast<const f32x4*>(p.bias + gn);
                    bias8[1] = *reinterpret_cast<const f32x4*>(p.bias + gn + 4);
                }
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int slot = mt % DEPTH;
        // this piece's operands move to `cur`, their slot is refilled right away: the loads run under this piece's work
        f32x4 rcur[4];
        i32x4 hcur[2][2];
        if constexpr (GLU == 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) { hcur[half][0] = hbuf[slot][half][0]; hcur[half][1] = hbuf[slot][half][1]; }
            if (mt + DEPTH < TM) load_h(mt + DEPTH, hbuf[slot]);
        }
        if (pre_res) {
#pragma unroll
            for (int i = 0; i < 4; ++i) rcur[i] = rbuf[slot][i];
            if (mt + DEPTH < TM) load_res(mt + DEPTH, rbuf[slot]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) patch[pidx(4 * g + r, 16 * nt + li)] = acc[mt][nt][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int rbase = row0 + 16 * mt;
        if constexpr (GLU == 1) {
            // fused SwiGLU forward (transformer.py:216-219): patch cols 0-31 = x, 32-63 = gate of act columns col0..+31
            const int r = lane >> 2, c8 = (lane & 3) * 8;
            const int gm = rbase + r, j0 = col0 + c8;
            if (gm < p.M && j0 < p.glu_inner) {
                float xv[8], gv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[e] = patch[pidx(r, c8 + e)]; gv[e] = patch[pidx(r, 32 + c8 + e)]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) { xv[e] += bx[e]; gv[e] += bg[e]; }
                i32x4 hx, hg, av;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hx[e] = (int)pack_bf16x2(xv[2 * e], xv[2 * e + 1]);
                    hg[e] = (int)pack_bf16x2(gv[2 * e], gv[2 * e + 1]);
                    // activation from the bf16-rounded pre-activation, exactly what the unfused kernel would read back
                    const float x0 = bf16lo((uint32_t)hx[e]), x1 = bf16hi((uint32_t)hx[e]);
                    const float g0 = bf16lo((uint32_t)hg[e]), g1 = bf16hi((uint32_t)hg[e]);
                    av[e] = (int)pack_bf16x2(x0 * siluf_(g0), x1 * siluf_(g1));
                }
                bf16_t* hp = reinterpret_cast<bf16_t*>(p.C) + (int64_t)gm * p.ldc;
                *reinterpret_cast<i32x4*>(hp + j0) = hx;
                *reinterpret_cast<i32x4*>(hp + p.glu_inner + j0) = hg;
                *reinterpret_cast<i32x4*>(static_cast<bf16_t*>(p.glu_aux) + (int64_t)gm * p.glu_inner + j0) = av;
            }
        } else if constexpr (GLU == 2) {
            // fused SwiGLU backward: acc = d(act)[m][j]; reads h, writes dh = (dact*silu(g), dact*x*silu'(g)), sums db.
            // (Round 3 tried the arithmetic on f32x2 pairs - v_pk_mul / v_pk_fma / v_pk_add_f32, one v_exp + v_rcp per element:
            // 21.4 -> 23.1 us per tile, and no longer bit-identical to the unfused kernel.  The VALU work is not what this
            // epilogue waits for; see the knock-out timings in DESIGN.md 5.3.)
            const int c8 = (lane & 7) * 8, j0 = col0 + c8;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = (lane >> 3) + 8 * half;
                const int gm = rbase + r;
                if (gm < p.M && j0 < p.glu_inner) {
                    const i32x4 xv = hcur[half][0], gv = hcur[half][1];
                    i32x4 ox, og;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x[2] = {bf16lo((uint32_t)xv[e]), bf16hi((uint32_t)xv[e])};
                        const float gg[2] = {bf16lo((uint32_t)gv[e]), bf16hi((uint32_t)gv[e])};
                        // the unfused path rounds dact to bf16 between the GEMM and the activation backward
                        const float d[2] = {bf16_to_f32(f32_to_bf16(patch[pidx(r, c8 + 2 * e)] * p.alpha)),
                                            bf16_to_f32(f32_to_bf16(patch[pidx(r, c8 + 2 * e + 1)] * p.alpha))};
                        float dx[2], dg[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const float sgm = sigmoidf_(gg[q]);
                            dx[q] = d[q] * gg[q] * sgm;
                            dg[q] = d[q] * x[q] * sgm * (1.f + gg[q] * (1.f - sgm));
                        }
                        ox[e] = (int)pack_bf16x2(dx[0], dx[1]);
                        og[e] = (int)pack_bf16x2(dg[0], dg[1]);
                        sx[2 * e] += bf16lo((uint32_t)ox[e]); sx[2 * e + 1] += bf16hi((uint32_t)ox[e]);
                        sg[2 * e] += bf16lo((uint32_t)og[e]); sg[2 * e + 1] += bf16hi((uint32_t)og[e]);
                    }
                    if (!(p.dbg & 2)) {
                        bf16_t* dp = reinterpret_cast<bf16_t*>(p.C) + (int64_t)gm * p.ldc;
                        *reinterpret_cast<i32x4*>(dp + j0) = ox;
                        *reinterpret_cast<i32x4*>(dp + p.glu_inner + j0) = og;
                    }
                }
            }
        } else if (p.atomic) {
            // split-K: 64 consecutive floats of one row per wave-instruction (256 contiguous bytes per atomic)
            const int gn = col0 + lane;
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                const int gm = rbase + r;
                if (gm < p.M && gn < p.N)
                    atomicAdd(reinterpret_cast<float*>(p.C) + (int64_t)gm * p.ldc + gn, patch[pidx(r, lane)] * p.alpha);
            }
        } else if constexpr (C_F32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = g + 4 * i;
                const int gm = rbase + r, gn = col0 + 4 * li;
                if (gm >= p.M || gn >= p.N) continue;
                const f32x4 pv = *reinterpret_cast<const f32x4*>(patch + pidx(r, 4 * li));
                float v[4] = {pv[0], pv[1], pv[2], pv[3]};
                const int64_t crow = gemm_crow(p, gm);
                gemm_epilogue4(p, gm, gn, crow, v, false, &bias4);
                if (pre_res) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += rcur[i][j];
                }
                float* cp = reinterpret_cast<float*>(p.C) + crow * p.ldc + gn;
                if (p.accumulate) {
                    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += c0[j];
                }
                *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
            }
        } else {
            // bf16 output: a lane owns 8 consecutive columns of a row -> ONE 16-byte store (8 lanes = a row's 128 bytes)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int r = (lane >> 3) + 8 * half;
                const int gm = rbase + r, gn = col0 + 8 * (lane & 7);
                if (gm >= p.M || gn >= p.N) continue;
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(patch + pidx(r, 8 * (lane & 7)));
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(patch + pidx(r, 8 * (lane & 7) + 4));
                float v0[4] = {p0[0], p0[1], p0[2], p0[3]}, v1[4] = {p1[0], p1[1], p1[2], p1[3]};
                const int64_t crow = gemm_crow(p, gm);
                gemm_epilogue4(p, gm, gn, crow, v0, true, &bias8[0]);
                gemm_epilogue4(p, gm, gn + 4, crow, v1, true, &bias8[1]);
                i32x4 o;
                o[0] = (int)pack_bf16x2(v0[0], v0[1]);
                o[1] = (int)pack_bf16x2(v0[2], v0[3]);
                o[2] = (int)pack_bf16x2(v1[0], v1[1]);
                o[3] = (int)pack_bf16x2(v1[2], v1[3]);
                *reinterpret_cast<i32x4*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gn) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch reads done before the next piece overwrites it
    }
    if constexpr (GLU == 2) {
        if (p.glu_dbias) {
            // lanes with equal (lane & 7) own the same 8 columns: fold the 8 row-lanes, then 16 atomics from lanes 0-7
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { sx[e] += __shfl_xor(sx[e], o, 64); sg[e] += __shfl_xor(sg[e], o, 64); }
            }
            const int j0 = col0 + (lane & 7) * 8;
            if (lane < 8 && j0 < p.glu_inner) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    atomicAdd(p.glu_dbias + j0 + e, sx[e]);
                    atomicAdd(p.glu_dbias + p.glu_inner + j0 + e, sg[e]);
                }
            }
        }
    }
    if (p.stamps) {
        gemm_stamp(p, wave, lane, 4, stamp_slot);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        gemm_stamp(p, wave, lane, 5, stamp_slot);
    }
}

#define MFMA16(FA, FB)                                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)          \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, FA[mt]),              \
                                                              __builtin_bit_cast(bf16x8, FB[nt]), acc[mt][nt], 0, 0, 0)

template <bool A_KM, bool B_KM, bool C_F32, int WM, int WN, int TM, int GLU = 0, int NS = 3>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm2_kernel(GemmParams p) {
    static_assert(TM == 4, "wave tile is 64 x 64");
    constexpr int TN = 4;
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    int tm, tn;
    gemm_tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk_all = (p.K + BK2 - 1) / BK2;
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int nk = min(p.ktiles_per_split, nk_all - kt0);
    const int kvalid = p.K % BK2;                                   // ragged last K-tile (0: none)
    const int tail_t = kvalid ? nk_all - 1 - kt0 : -1;              // its index inside this workgroup's K range

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Loader<A_KM, BM, NW> la;
    Loader<B_KM, BN, NW> lb;
    la.init(p.A, p.lda, p.M, m0, kt0 * BK2, wave, lane);
    lb.init(p.B, p.ldb, p.N, n0, kt0 * BK2, wave, lane, GLU == 1 ? p.glu_inner : 0, tn);
    const int arow = wm * TM * 16, bcol = wn * TN * 16;
    Reader<A_KM, BM> ra;
    Reader<B_KM, BN> rb;
    ra.init(lds0, arow, lane);
    rb.init(lds0 + A_BYTES, bcol, lane);

    {
    // prologue: up to NS tiles in flight (ring of NS stages); wait for tile 0 (for tile 1 too where the 6-bit vmcnt could not
    // name NS - 1 tiles)
    constexpr int PT = Loader<A_KM, BM, NW>::PER_WAVE + Loader<B_KM, BN, NW>::PER_WAVE;   // DMA pieces per tile per wave
    static_assert((NS - 2) * PT <= 63, "vmcnt is a 6-bit counter");
    {
        const int pre = nk < NS ? nk : NS;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (t < pre) {
                la.issue_at(smem + t * STAGE, wave, lane, t, tail_t, kvalid);
                lb.issue_at(smem + t * STAGE + A_BYTES, wave, lane, t, tail_t, kvalid);
            }
        }
        constexpr int PRE_KEEP = (NS - 1) * PT <= 63 ? NS - 1 : NS - 2;
        const int keep = pre - 1 < PRE_KEEP ? pre - 1 : PRE_KEEP;   // tiles that may stay in flight
        // (a compile-time immediate per case: the chain below is a few compares, executed once)
        auto wait_keep = [&](auto self, auto c) {
            constexpr int C = decltype(c)::value;
            if (keep == C) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C * PT) : "memory");
            if constexpr (C > 0) self(self, std::integral_constant<int, C - 1>{});
        };
        wait_keep(wait_keep, std::integral_constant<int, PRE_KEEP>{});
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    i32x4 fa0[4], fb0[4], fa1[4], fb1[4];
    ra.template read<0, 4>(0, fa0);
    rb.template read<0, 4>(0, fb0);

    unsigned so_cur = 0, so_nxt = STAGE;   // byte offsets of the stage being computed / the next one (ring of NS)
    int kt = 0;

    // one K-tile: ISSUE = start the DMA of tile kt+NS, NEXT = a tile kt+1 exists, KEEP = tiles (kt+2 ...) whose DMA stays in flight
    auto iteration = [&](auto issue_c, auto next_c, auto keep_c) {
        constexpr bool ISSUE = decltype(issue_c)::value, NEXT = decltype(next_c)::value;
        constexpr int KEEP = decltype(keep_c)::value;
        // ---- k-step 0: set 0 has landed (reads issued one phase ago); read set 1 while the MFMAs of set 0 issue
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ra.template read<1, 4>(so_cur, fa1);
        rb.template read<1, 4>(so_cur, fb1);
        __builtin_amdgcn_sched_barrier(0);
        MFMA16(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hand-over: my reads of this stage are done; everybody's DMA of tile kt+1 has landed; the DMA of the tiles behind
        //      it (issued in earlier iterations) stays in flight across the barrier: counted vmcnt, never drained
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP * PT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // Stagger (MI355X_MICROARCH "two waves per SIMD", item 9): the two waves of a SIMD (w and w+4) would otherwise
        // run this block in lockstep - both issuing DMA/LDS reads, then both queueing on the matrix pipe.  The
        // second-dispatched half does its MFMAs first and its loads second, so one partner computes while the other loads.
        if (wave >= NW / 2) {
            MFMA16(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (ISSUE) {      // this stage is free: start the DMA of tile kt+NS into it
            la.issue_at(smem + so_cur, wave, lane, kt + NS, tail_t, kvalid);
            lb.issue_at(smem + so_cur + A_BYTES, wave, lane, kt + NS, tail_t, kvalid);
        }
        // ---- k-step 1: read set 0 of the next tile while the MFMAs of set 1 issue
        if constexpr (NEXT) {
            ra.template read<0, 4>(so_nxt, fa0);
            rb.template read<0, 4>(so_nxt, fb0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wave < NW / 2) {
            MFMA16(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        so_cur = so_nxt;
        so_nxt = so_nxt + STAGE >= NS * STAGE ? 0 : so_nxt + STAGE;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
#pragma unroll 1
    for (; kt + NS < nk; ++kt) iteration(T_{}, T_{}, std::integral_constant<int, NS - 2>{});
    // tail: R tiles follow the current one (R = NS - 1 ... 1), R - 1 of them may stay in flight; then the last tile
    auto tail = [&](auto self, auto r) {
        constexpr int R = decltype(r)::value;
        if (nk - 1 - kt == R) { iteration(F_{}, T_{}, std::integral_constant<int, R - 1>{}); ++kt; }
        if constexpr (R > 1) self(self, std::integral_constant<int, R - 1>{});
    };
    tail(tail, std::integral_constant<int, NS - 1>{});
    iteration(F_{}, F_{}, std::integral_constant<int, 0>{});
    }

    if constexpr (C_F32) {
        if (p.slab_stride) {        // split-K into per-slice slabs (few-rows path)
            GemmParams q = p;
            q.C = static_cast<float*>(p.C) + (int64_t)blockIdx.y * p.slab_stride;
            wave_epilogue<C_F32, TM>(q, acc, smem, wave, lane, m0 + arow, n0 + bcol);
            return;
        }
    }
    // (fused SwiGLU forward: a wave's 64 tile columns are 32 x columns + the 32 matching gate columns, see Loader::init)
    wave_epilogue<C_F32, TM, GLU>(p, acc, smem, wave, lane, m0 + arow, GLU == 1 ? tn * (BN / 2) + wn * 32 : n0 + bcol);
}


// ---- the same tile with TWO wave groups (few-row GEMMs: one workgroup per CU at best, so nobody covers a wave's LDS / DMA-issue
// latency but its own SIMD partner): group 0 multiplies the first 32-deep k-step of every stage, group 1 the second - two waves per
// SIMD on the same 64 x 64 sub-tile, 16 MFMAs per wave per K-tile, the DMA pieces spread over twice the waves; group 1's partial
// sums travel through LDS to group 0 before the epilogue.  Sum order per output: (k-step 0 of every tile) + (k-step 1 of every
// tile) - fixed, so results are reproducible, but not bit-identical to the one-group kernel.
template <bool B_KM, bool C_F32, int WM, int WN, int GLU, int NS>
__global__ __launch_bounds__(WM * WN * 128, 1) void gemm2_ks2_kernel(GemmParams p) {
    static_assert(!B_KM || WN >= 2, "a k-major operand tile needs >= 128 columns (16 swizzle slots per k-row)");
    constexpr int TM = 4, TN = 4;
    constexpr int NWT = WM * WN, NW = 2 * NWT;
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int RED_OFF = 32768;                                   // behind the epilogue's wave patches
    static_assert(RED_OFF + NWT * 16384 <= NS * STAGE, "partial-sum exchange must fit into the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NWT, wt = wave % NWT;
    const int wm = wt / WN, wn = wt % WN;

    int tm, tn;
    gemm_tile_coords(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk_all = (p.K + BK2 - 1) / BK2;
    const int kt0 = blockIdx.y * p.ktiles_per_split;
    const int nk = min(p.ktiles_per_split, nk_all - kt0);
    const int kvalid = p.K % BK2;
    const int tail_t = kvalid ? nk_all - 1 - kt0 : -1;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Loader<false, BM, NW> la;
    Loader<B_KM, BN, NW> lb;
    la.init(p.A, p.lda, p.M, m0, kt0 * BK2, wave, lane);
    lb.init(p.B, p.ldb, p.N, n0, kt0 * BK2, wave, lane, GLU == 1 ? p.glu_inner : 0, tn);
    const int arow = wm * 64, bcol = wn * 64;
    Reader<false, BM> ra;
    Reader<B_KM, BN> rb;
    ra.init(lds0, arow, lane);
    rb.init(lds0 + A_BYTES, bcol, lane);

    constexpr int PT = Loader<false, BM, NW>::PER_WAVE + Loader<B_KM, BN, NW>::PER_WAVE;
    static_assert((NS - 1) * PT <= 63, "vmcnt is a 6-bit counter");
    {
        const int pre = nk < NS ? nk : NS;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            if (t < pre) {
                la.issue_at(smem + t * STAGE, wave, lane, t, tail_t, kvalid);
                lb.issue_at(smem + t * STAGE + A_BYTES, wave, lane, t, tail_t, kvalid);
            }
        }
        const int keep = pre - 1;
        auto wait_keep = [&](auto self, auto c) {
            constexpr int C = decltype(c)::value;
            if (keep == C) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C * PT) : "memory");
            if constexpr (C > 0) self(self, std::integral_constant<int, C - 1>{});
        };
        wait_keep(wait_keep, std::integral_constant<int, NS - 1>{});
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    i32x4 fa[2][4], fb[2][4];            // my k-step's fragments of the tile being multiplied / of the next one
    auto rd = [&](unsigned so, i32x4 (&A)[4], i32x4 (&B)[4]) {
        if (kg == 0) { ra.template read<0, 4>(so, A); rb.template read<0, 4>(so, B); }
        else { ra.template read<1, 4>(so, A); rb.template read<1, 4>(so, B); }
    };
    rd(0, fa[0], fb[0]);
    unsigned so_cur = 0, so_nxt = STAGE;
    int kt = 0;

    // one K-tile; P = which fragment buffer holds it; KEEP = tiles (kt+2 ...) whose DMA may stay in flight at the hand-over
    auto iteration = [&](auto par_c, auto issue_c, auto next_c, auto keep_c) {
        constexpr int P = decltype(par_c)::value, KEEP = decltype(keep_c)::value;
        constexpr bool ISSUE = decltype(issue_c)::value, NEXT = decltype(next_c)::value;
        // my fragments of tile kt are in registers (so stage kt may be refilled once everybody says the same) and my pieces of
        // tile kt+1 have landed
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP * PT) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) rd(so_nxt, fa[1 - P], fb[1 - P]);
        __builtin_amdgcn_sched_barrier(0);
        MFMA16(fa[P], fb[P]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ISSUE) {
            la.issue_at(smem + so_cur, wave, lane, kt + NS, tail_t, kvalid);
            lb.issue_at(smem + so_cur + A_BYTES, wave, lane, kt + NS, tail_t, kvalid);
        }
        so_cur = so_nxt;
        so_nxt = so_nxt + STAGE >= NS * STAGE ? 0 : so_nxt + STAGE;
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using KS_ = std::integral_constant<int, NS - 2>;
#pragma unroll 1
    for (; kt + NS + 1 < nk; kt += 2) {
        iteration(P0{}, T_{}, T_{}, KS_{});
        ++kt; iteration(P1{}, T_{}, T_{}, KS_{}); --kt;
    }
    int par = 0;
    if (kt + NS < nk) { iteration(P0{}, T_{}, T_{}, KS_{}); ++kt; par = 1; }
    auto tail = [&](auto self, auto r) {
        constexpr int R = decltype(r)::value;
        if (nk - 1 - kt == R) {
            if (par) iteration(P1{}, F_{}, T_{}, std::integral_constant<int, R - 1>{});
            else iteration(P0{}, F_{}, T_{}, std::integral_constant<int, R - 1>{});
            ++kt; par ^= 1;
        }
        if constexpr (R > 1) self(self, std::integral_constant<int, R - 1>{});
    };
    tail(tail, std::integral_constant<int, NS - 1>{});
    if (par) iteration(P1{}, F_{}, F_{}, P0{});
    else iteration(P0{}, F_{}, F_{}, P0{});

    // ---- group 1 hands its partial sums to group 0 (register layout kept: 16 bytes per lane per accumulator, conflict-free)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = reinterpret_cast<float*>(smem + RED_OFF) + wt * 4096;
    if (kg == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(red + ((i * TN + j) * 64 + lane) * 4) = acc[i][j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kg == 1) {
        __builtin_amdgcn_s_barrier();       // (the barrier at the top of wave_epilogue, which group 0 is entering)
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(red + ((i * TN + j) * 64 + lane) * 4);
    if constexpr (C_F32) {
        if (p.slab_stride) {
            GemmParams q = p;
            q.C = static_cast<float*>(p.C) + (int64_t)blockIdx.y * p.slab_stride;
            wave_epilogue<C_F32, TM>(q, acc, smem, wt, lane, m0 + arow, n0 + bcol);
            return;
        }
    }
    wave_epilogue<C_F32, TM, GLU>(p, acc, smem, wt, lane, m0 + arow, GLU == 1 ? tn * (BN / 2) + wn * 32 : n0 + bcol);
}

template <bool B_KM, bool C_F32, int WM, int WN, int GLU, int NS>
int launch2_ks2(const GemmParams& p, hipStream_t st) {
    constexpr int lds = NS * (WM + WN) * 64 * 128;
    static_assert(lds <= 160 * 1024, "LDS per workgroup");
    static std::atomic<uint64_t> lds_ok{0};
    kalle_allow_lds(reinterpret_cast<const void*>(gemm2_ks2_kernel<B_KM, C_F32, WM, WN, GLU, NS>), lds, lds_ok);
    dim3 grid(p.tiles_m * p.tiles_n, p.splits), block(WM * WN * 128);
    KALLE_LAUNCH((gemm2_ks2_kernel<B_KM, C_F32, WM, WN, GLU, NS>), grid, block, lds, st, p);
    return kalle_check_launch();
}

template <bool A_KM, bool B_KM, bool C_F32, int WM, int WN, int TM, int GLU = 0, int NS = 3>
int launch2(const GemmParams& p, hipStream_t st) {
    constexpr int BM = WM * TM * 16, BN = WN * 64;
    constexpr int lds = NS * (BM + BN) * 128;
    static_assert(lds <= 160 * 1024, "LDS per workgroup");
    static std::atomic<uint64_t> lds_ok{0};
    kalle_allow_lds(reinterpret_cast<const void*>(gemm2_kernel<A_KM, B_KM, C_F32, WM, WN, TM, GLU, NS>), lds, lds_ok);
    dim3 grid(p.tiles_m * p.tiles_n, p.splits), block(WM * WN * 64);
    KALLE_LAUNCH((gemm2_kernel<A_KM, B_KM, C_F32, WM, WN, TM, GLU, NS>), grid, block, lds, st, p);
    return kalle_check_launch();
}


// ================================================================================================ 256 x 256 tile
// 8 waves as 2 (M) x 4 (N), 128 x 64 per wave (128 accumulator VGPRs), two 64-KiB LDS stages, ONE fragment set:
// a wave reads the 12 fragments of a 32-deep k-step, waits, issues its 32 MFMAs; latency is covered by its SIMD
// partner, which runs half a phase apart (second half of the workgroup defers the last MFMA block of a K-tile past
// the barrier).  Per MFMA this tile needs 25 % fewer LDS reads, 33 % fewer DMA pieces and L2->LDS bytes than 256x128.
// build switches: KALLE_GEMM_PIPE = 1 software-pipelined main loop (fragments refilled row by row; measured 3-4 % SLOWER over the
// train step, 226.9 -> 235.4 ms same box: the loop is not waiting for LDS latency - on all-zero operands, i.e. at full clock, every
// shape runs in the time of its LDS-DMA stream alone, on random operands the chip holds ~2.0 GHz - and spreading the DMA issue
// over the second k-step halves the time the pieces have to land), 0 = the round-1 loop (default: whole k-step read in a burst,
// SIMD partners half a phase apart, all DMA of K-tile kt + 2 issued right behind the hand-over); KALLE_GEMM_KNOCKOUT (variant builds for knock-out timing of the main
// loop, tools/gemm_stamps.py): 1 no LDS-DMA after the prologue, 2 no fragment reads, 3 no MFMAs, 4 neither reads nor MFMAs
#ifndef KALLE_GEMM_PIPE
#define KALLE_GEMM_PIPE 0
#endif
#ifndef KALLE_GEMM_KNOCKOUT
#define KALLE_GEMM_KNOCKOUT 0
#endif
#define KO_READ(...) do { if (KALLE_GEMM_KNOCKOUT != 2 && KALLE_GEMM_KNOCKOUT != 4) { __VA_ARGS__; } } while (0)
#define KO_MFMA(...) do { if (KALLE_GEMM_KNOCKOUT != 3 && KALLE_GEMM_KNOCKOUT != 4) { __VA_ARGS__; } } while (0)
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

constexpr int GEMM3_LDS = 2 * (256 + 256) * 128 + 8 * 16 * 64 * 4;      // two 64-KiB stages + the epilogue patches = 160 KiB

// compute units of the current device (cached per device ordinal)
static int kalle_cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cached[dev & 63].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    v &= ~7;                                // a multiple of 8: a workgroup's tiles stay on one XCD's run of the raster
    if (v < 8) v = 8;
    cached[dev & 63].store(v, std::memory_order_relaxed);
    return v;
}

#define MFMA32(FA, FB)                                                                                         \
    _Pragma("unroll") for (int mt = 0; mt < 8; ++mt) _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)          \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, FA[mt]),              \
                                                              __builtin_bit_cast(bf16x8, FB[nt]), acc[mt][nt], 0, 0, 0)

// ---- dynamic tile hand-out of the persistent kernel ---------------------------------------------------------------------------
// With a static walk (tiles b, b + grid, ...) a workgroup that starts late does its whole share late: when another kernel holds
// some CUs while the GEMM launches - RCCL's all-reduce of the previous block's gradient bucket on the communication stream, the
// optimizer slice on its side stream; neither can share a CU with a 160-KiB-LDS workgroup - the workgroups that found no CU run
// after the others have finished (tools/cu_hold_probe.py: 16-64 CUs held -> 1.25-1.7 x the GEMM's time).  Instead every
// workgroup takes tile b first and then DRAWS: counter x (one per XCD, each in its own 128-byte line, so that a workgroup keeps
// walking the XCD-contiguous run of the raster whose panels its L2 holds) hands out ids x + 8 (grid / 8 + j) to the workgroups
// of XCD x.  The draw is one returning atomic per tile by one lane and runs TWO
// tiles ahead, because every K-tile of the main loop ends in s_waitcnt vmcnt(0) and would wait for it: it is issued at the start
// of tile i's epilogue (for tile i + 2), right behind the requests for tile i + 1's first K-tiles, and its round trip runs under
// the epilogue; at the start of tile i + 1 lane 0 leaves the id in wave 0's (idle) epilogue patch, the tile's first barrier
// publishes it and every wave keeps it in an SGPR until the epilogue's hook asks which tile to prefetch.  Counter sets are self-cleaning: the
// last workgroup to leave zeroes the set.
constexpr int SCHED_SETS = 256;
constexpr int SCHED_LINE = 32;                                   // ints per counter (one 128-byte line each)
__device__ int kalle_gemm_sched[SCHED_SETS][9 * SCHED_LINE];     // [set]: 8 per-XCD counters, then the count of workgroups that left

// the draw is split so that nothing waits for the atomic where it is issued: sched_issue returns the raw counter value (in
// flight), sched_resolve - a tile later - turns it into a tile id.  There is no stealing across XCDs: a first version walked the
// other XCDs' counters once a workgroup's own run was exhausted - every workgroup did that once per kernel, at the start of its
// last tile, seven blocking round trips for nothing in the undisturbed case (+8 % on a 150-us GEMM) - and kernels that hold CUs
// are spread over the XCDs by the dispatcher like everything else.
__device__ __forceinline__ int sched_issue(int* sched, int xcd) { return atomicAdd(sched + xcd * SCHED_LINE, 1); }
__device__ __forceinline__ int sched_resolve(int xcd, int raw, int first_round, int ntiles) {
    const int id = xcd + 8 * (first_round + raw);
    return id < ntiles ? id : ntiles;
}
__device__ __forceinline__ void sched_leave(int* sched, int nwg) {
    // (no fence: every draw this workgroup issued has returned - its value was used - and the counters are only ever touched by
    // device-scope atomics; an agent-scope release here would write back the XCD's L2, 256 times per kernel)
    if (atomicAdd(sched + 8 * SCHED_LINE, 1) == nwg - 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicExch(sched + i * SCHED_LINE, 0);
    }
}

// 256 x 256 output tiles `tile0, tile0 + tstride, ...` (< tile_end; ids in the XCD-aware grouped raster, or plain raster for the
// grouped launch) over K-tiles [kt0, kt0 + nk) of problem `p` - shared by the plain, the persistent and the grouped kernel.
// PERSISTENT form (tstride > 0, one workgroup per CU walks its tiles): what the in-kernel stamps showed per K = 1536 tile - 1.6-2.4 us
// from entry to the first K-tile landed, ~1 us between a workgroup's exit and its successor's entry - is taken off the critical
// path: the next tile's K-tile 0 is requested at the START of the last K-tile of the current one (into the stage that fell free
// at the previous hand-over), its K-tile 1 right after the epilogue's barrier (into the stage of the last K-tile), and the
// epilogue's LDS patch lives in the 32 KiB the two stages leave free.  The stores of the epilogue are younger than those DMA
// requests in the wave's VMEM queue, so the next tile starts with vmcnt(0) (stores are acknowledged ~0.2 us after issue).
template <bool A_KM, bool B_KM, bool C_F32, int GLU, bool XCD_RASTER, bool PERSIST>
__device__ __forceinline__ void gemm3_tiles(const GemmParams& p, int tile0, int tstride, int tile_end, int ntiles_raster, int kt0,
                                            int nk) {
    const int kvalid = p.K % BK2;                                               // ragged last K-tile (0: none)
    const int tail_t = kvalid ? (p.K + BK2 - 1) / BK2 - 1 - kt0 : -1;           // its index inside [0, nk)
    constexpr int WN = 4, TM = 8, NW = 8;
    constexpr int BM = 256, BN = 256;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)LDS_PTR(char, smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int arow = wm * 128, bcol = wn * 64;
    Loader<A_KM, BM, NW> la;
    Loader<B_KM, BN, NW> lb;
    Reader<A_KM, BM> ra;
    Reader<B_KM, BN> rb;
    ra.init(lds0, arow, lane);
    rb.init(lds0 + A_BYTES, bcol, lane);
    const bool late = wave >= NW / 2;     // the staggered half
#if !KALLE_GEMM_PIPE
    if (late) __builtin_amdgcn_s_setprio(1);   // the second-dispatched half loses every issue arbitration by age otherwise
#endif

    auto coords = [&](int tile, int& tm, int& tn) {
        if constexpr (XCD_RASTER) gemm_tile_coords(tile, ntiles_raster, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
        else gemm_tile_coords_plain(tile, p.tiles_m, p.tiles_n, p.group_m, tm, tn);
    };
    auto init_loaders = [&](int tm, int tn, int kt_first) {      // source pointers of K-tile `kt_first` of this workgroup's range
        // (KALLE_GEMM_DBG bit 2, timing experiment only - results are wrong: every workgroup fetches the operand panels of tile
        // (tm & 1, tn & 1), i.e. all of an XCD's workgroups share four panels and the L2 -> LDS stream runs at ~100 % L2 hits)
        if (p.dbg & 4) { tm &= 1; tn &= 1; }
        la.init(p.A, p.lda, p.M, tm * BM, (kt0 + kt_first) * BK2, wave, lane);
        lb.init(p.B, p.ldb, p.N, tn * BN, (kt0 + kt_first) * BK2, wave, lane, GLU == 1 ? p.glu_inner : 0, tn);
    };
    if constexpr (PERSIST) {
        if (p.dephase_ticks > 0 && ((blockIdx.x >> 3) & 1)) {
            const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (unsigned long long)p.dephase_ticks;
            while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(32);
        }
    }
    bool prefetched = false;              // K-tiles 0 (and 1) of `tile` were requested by the previous tile's tail
    unsigned s0 = 0;                      // stage (byte offset) that holds K-tile 0 of `tile`

    // (the software-pipelined build variant keeps the static walk)
    int* const sched = PERSIST && !KALLE_GEMM_PIPE && p.sched_set > 0 ? &kalle_gemm_sched[p.sched_set - 1][0] : nullptr;
    int next_tile = tile_end;
    int drawn = 0;                        // lane 0 of wave 0: the counter value drawn last (two tiles ahead of the one being computed)
    if constexpr (PERSIST) {
        if (sched && tid == 0) drawn = sched_issue(sched, blockIdx.x & 7);                             // the tile after the first
    }
    volatile int* const mailbox = reinterpret_cast<volatile int*>(smem + 2 * STAGE);   // wave 0's epilogue patch (idle in the main loop)
    for (int tile = tile0; tile < tile_end; tile = next_tile) {
        int tm, tn;
        coords(tile, tm, tn);
        const int m0 = tm * BM, n0 = tn * BN;
        next_tile = PERSIST ? tile + tstride : tile_end;       // (dynamic hand-out: read from the mailbox behind the first barrier)
        bool has_next = PERSIST && next_tile < tile_end;
        gemm_stamp(p, wave, lane, 0, tile);
        f32x4 acc[TM][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        if (!prefetched) {
            init_loaders(tm, tn, 0);
            la.issue_at(smem, wave, lane, 0, tail_t, kvalid);
            lb.issue_at(smem + A_BYTES, wave, lane, 0, tail_t, kvalid);
            s0 = 0;
            if (nk > 1) {
                la.issue_at(smem + STAGE, wave, lane, 1, tail_t, kvalid);
                lb.issue_at(smem + STAGE + A_BYTES, wave, lane, 1, tail_t, kvalid);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // 4 + 4 DMA pieces per tile per wave
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if constexpr (PERSIST) {
            init_loaders(tm, tn, nk > 1 ? 2 : 1);                  // (K-tiles 0 and 1 are on their way)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // both prefetched K-tiles + the previous epilogue's stores
        }
        if constexpr (PERSIST) {
            if (sched && tid == 0) {          // (the draw has returned: the waits above cover it)
                *mailbox = sched_resolve(blockIdx.x & 7, drawn, tstride >> 3, tile_end);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PERSIST) {
            if (sched) {
                next_tile = __builtin_amdgcn_readfirstlane(*mailbox);
                has_next = next_tile < tile_end;
            }
        }
        gemm_stamp(p, wave, lane, 1, tile);

#if KALLE_GEMM_PIPE
        // ---- software-pipelined main loop (round 3): every wave keeps its own MFMA stream fed.  A k-step is 8 rows of 4 MFMAs
        // (acc[r][0..3] += A_r x B_0..3); right behind row r the register of A_r is refilled with the NEXT k-step's A_r, and rows
        // 0-3 also fetch the next k-step's B_r into the second B set - one or two LDS reads per 4 MFMAs instead of 12-24 reads in
        // a burst that only the SIMD partner could cover; the 8 DMA pieces of K-tile kt + 2 go out one per row of the second
        // k-step.  LDS reads return in order, so the waits are counted: before row 0 everything but the four youngest A
        // fragments, before row r >= 1 everything up to A_r (= all but 7 A + 4 B fragments' worth of instructions; the counter
        // has 4 bits).  One barrier per K-tile, at the k-step boundary: behind it the stage just read is free for the DMA of
        // K-tile kt + 2 and the other stage (K-tile kt + 1, requested a K-tile ago) may be read.
        i32x4 fa[8], fb[2][4];
        constexpr int NIA = Reader<A_KM, BM>::NI, NIB = Reader<B_KM, BN>::NI;
        constexpr int W0 = 4 * NIA < 15 ? 4 * NIA : 15;
        constexpr int WR = 7 * NIA + 4 * NIB < 15 ? 7 * NIA + 4 * NIB : 15;
        unsigned so_cur = s0;
        int kt = 0;
        // CUR: B set multiplied; RS: k-step whose fragments are fetched behind the rows (from stage so_rd); READ / WAIT / ISSUE:
        // fetch at all / counted waits in front of the rows / DMA pieces of K-tile `kt_dma` into stage so_dma
        auto kstep = [&](auto cur_c, auto rs_c, auto read_c, auto wait_c, auto issue_c, unsigned so_rd, unsigned so_dma, bool dma_tail) {
            constexpr int CUR = decltype(cur_c)::value, RS = decltype(rs_c)::value;
            constexpr bool READ = decltype(read_c)::value, WAIT = decltype(wait_c)::value;
            constexpr bool ISSUE = decltype(issue_c)::value && KALLE_GEMM_KNOCKOUT != 1;
            static_for<8>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (WAIT) {
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(r == 0 ? W0 : WR) : "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    KO_MFMA(acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[r]),
                                                                                __builtin_bit_cast(bf16x8, fb[CUR][nt]), acc[r][nt], 0, 0, 0));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ISSUE) {
                    if constexpr (r < 4) la.template issue_one<r>(smem + so_dma, wave, lane, dma_tail, kvalid);
                    else lb.template issue_one<r - 4>(smem + so_dma + A_BYTES, wave, lane, dma_tail, kvalid);
                }
                if constexpr (READ) {
                    KO_READ(ra.template read1<RS, r>(so_rd, fa[r]));
                    if constexpr (r < 4) KO_READ(rb.template read1<RS, r>(so_rd, fb[1 - CUR][r]));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // prologue: k-step 0 of K-tile 0, in the issue order the counted waits assume
        static_for<8>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            ra.template read1<0, r>(so_cur, fa[r]);
            if constexpr (r < 4) rb.template read1<0, r>(so_cur, fb[0][r]);
        });
        __builtin_amdgcn_sched_barrier(0);
        auto iteration = [&](auto issue_c) {
            kstep(I0{}, I1{}, T_{}, T_{}, F_{}, so_cur, 0u, false);
            // hand-over: my reads of this stage have returned, my DMA pieces of K-tile kt + 1 have landed - and everybody's
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            kstep(I1{}, I0{}, T_{}, F_{}, issue_c, so_cur ^ STAGE, so_cur, kt + 2 == tail_t);
            so_cur ^= STAGE;
        };
#pragma unroll 1
        for (; kt + 2 < nk; ++kt) iteration(T_{});
        if (kt + 1 < nk) { iteration(F_{}); ++kt; }
        // the last K-tile: nothing to hand over (the epilogue's barrier follows)
        kstep(I0{}, I1{}, T_{}, T_{}, F_{}, so_cur, 0u, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        kstep(I1{}, I0{}, F_{}, F_{}, F_{}, 0u, 0u, false);
#else
        i32x4 fa[8], fb[4];
        unsigned so_cur = s0;
        int kt = 0;
        // prologue of the software pipeline: k-step 0 of tile 0
        ra.template read<0, 8>(so_cur, fa);
        rb.template read<0, 4>(so_cur, fb);

        // (variant builds for knock-out timing of the main loop, tools/gemm_stamps.py: -DKALLE_GEMM_KNOCKOUT=1 no LDS-DMA after
        // the prologue, =2 no fragment reads, =3 no MFMAs, =4 neither reads nor MFMAs; the product build has none of it)
        auto iteration = [&](auto issue_c, auto next_c) {
            constexpr bool ISSUE = decltype(issue_c)::value && KALLE_GEMM_KNOCKOUT != 1, NEXT = decltype(next_c)::value;
            // k-step 0 of this tile is in flight / landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            KO_MFMA(MFMA32(fa, fb));
            __builtin_amdgcn_sched_barrier(0);
            KO_READ(ra.template read<1, 8>(so_cur, fa); rb.template read<1, 4>(so_cur, fb));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my last reads of this stage are done
            __builtin_amdgcn_sched_barrier(0);
            if (!late) {
                KO_MFMA(MFMA32(fa, fb));
                __builtin_amdgcn_sched_barrier(0);
            }
            // hand-over: everybody's DMA of tile kt+1 has landed, everybody is done reading this stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (late) {
                KO_MFMA(MFMA32(fa, fb));
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (ISSUE) {      // this stage is free: start the DMA of tile kt+2 into it
                la.issue_at(smem + so_cur, wave, lane, kt + 2, tail_t, kvalid);
                lb.issue_at(smem + so_cur + A_BYTES, wave, lane, kt + 2, tail_t, kvalid);
            }
            so_cur ^= STAGE;
            if constexpr (NEXT) {
                KO_READ(ra.template read<0, 8>(so_cur, fa); rb.template read<0, 4>(so_cur, fb));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // the last K-tile: no hand-over (the epilogue's barrier follows)
        auto last_iteration = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MFMA32(fa, fb);
            __builtin_amdgcn_sched_barrier(0);
            ra.template read<1, 8>(so_cur, fa);
            rb.template read<1, 4>(so_cur, fb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MFMA32(fa, fb);
            __builtin_amdgcn_sched_barrier(0);
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
#pragma unroll 1
        for (; kt + 2 < nk; ++kt) iteration(T_{}, T_{});
        if (kt + 1 < nk) { iteration(F_{}, T_{}); ++kt; }
        // (register allocation, not timing, picks the form: without the hand-over barrier the compiler starts the epilogue's set-up
        // under the last MFMAs - 27 more VGPRs in the one-tile kernels, 33 spilled in the fused SwiGLU backward)
        if constexpr (!PERSIST || GLU == 2) iteration(F_{}, F_{});
        else last_iteration();
#endif
        gemm_stamp(p, wave, lane, 2, tile);

        // Behind the epilogue's barrier both stages are free: request the next tile's K-tiles 0 and 1 there.  The loaders are
        // set up for these two requests only and die with them (kept alive across the epilogue they cost 16 VGPRs and spills);
        // the next tile re-initialises them two K-tiles further on.
        const unsigned s_last = so_cur;                 // stage of the last K-tile -> next K-tile 1; the other one -> next K-tile 0
        auto hook = [&]() {
            if constexpr (!PERSIST) return;
            if (has_next) {
                int ntm, ntn;
                coords(next_tile, ntm, ntn);
                init_loaders(ntm, ntn, 0);
                la.issue_at(smem + (s_last ^ STAGE), wave, lane, 0, tail_t, kvalid);
                lb.issue_at(smem + (s_last ^ STAGE) + A_BYTES, wave, lane, 0, tail_t, kvalid);
                if (nk > 1) {
                    la.issue_at(smem + s_last, wave, lane, 1, tail_t, kvalid);
                    lb.issue_at(smem + s_last + A_BYTES, wave, lane, 1, tail_t, kvalid);
                }
                // the tile after the next one: the atomic's round trip runs under the epilogue (it is older than the epilogue's
                // loads and stores, so nothing at the next tile's start waits for it)
                if (sched && tid == 0) drawn = sched_issue(sched, blockIdx.x & 7);
            }
        };
        wave_epilogue<C_F32, TM, GLU>(p, acc, smem + 2 * STAGE, wave, lane, m0 + arow, GLU == 1 ? tn * 128 + wn * 32 : n0 + bcol,
                                      hook, tile);
        prefetched = has_next;
        s0 = s_last ^ STAGE;
        if constexpr (!PERSIST) break;
    }
}

// Two forms, one per instantiation (both in one kernel cost 40 VGPRs and spills): the weight-gradient layout (k-major A) is the only
// one that is ever cut along K - one workgroup per (tile, K slice), uniform or mixed split; every other layout runs whole-K tiles
// on the persistent form.
template <bool A_KM, bool B_KM, bool C_F32, int GLU = 0>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(GemmParams p) {
    const int nk_all = (p.K + BK2 - 1) / BK2;
    const int ntiles = p.tiles_m * p.tiles_n;
    if constexpr (A_KM) {
        if (p.mix_na >= 0) {
            // mixed split-K: 1-D grid; blocks [0, na * sa) = tiles [0, na) x sa slices (tile fastest), then the other tiles x (sa + 1)
            const int na = p.mix_na, sa = p.mix_sa;
            int bid = blockIdx.x, tile, slice, per;
            if (bid < na * sa) { tile = bid % na; slice = bid / na; per = (nk_all + sa - 1) / sa; }
            else { bid -= na * sa; const int nb = ntiles - na; tile = na + bid % nb; slice = bid / nb; per = (nk_all + sa) / (sa + 1); }
            const int kt0 = slice * per, nk = min(per, nk_all - kt0);
            if (nk <= 0) return;                 // (uniform per workgroup)
            gemm3_tiles<A_KM, B_KM, C_F32, GLU, true, false>(p, tile, 0, tile + 1, ntiles, kt0, nk);
        } else {
            const int kt0 = blockIdx.y * p.ktiles_per_split;
            gemm3_tiles<A_KM, B_KM, C_F32, GLU, true, false>(p, blockIdx.x, 0, blockIdx.x + 1, ntiles, kt0,
                                                             min(p.ktiles_per_split, nk_all - kt0));
        }
    } else {
        // whole-K tiles: gridDim.x <= tiles workgroups (one per CU) walk tiles b, b + grid, ...; a multiple-of-8 grid keeps a
        // workgroup's tiles on the XCD-contiguous run of the raster its XCD owns
        gemm3_tiles<A_KM, B_KM, C_F32, GLU, true, true>(p, blockIdx.x, gridDim.x, ntiles, ntiles, 0, nk_all);
        if (!KALLE_GEMM_PIPE && p.sched_set > 0 && threadIdx.x == 0) sched_leave(&kalle_gemm_sched[p.sched_set - 1][0], gridDim.x);
    }
}

// ================================================================================================ grouped weight gradients
// All weight gradients of a transformer block in ONE launch (kalle_gemm_wgrad_group): dW_i[N_i, K_i] += dY_i^T X_i, every one a
// (k-major, k-major, fp32) problem over the block's tokens.  Launched one by one they need 3 - 8 K slices each to fill 256 CUs
// (36 - 288 output tiles) and pay for it in fp32 atomics (3.6 x the algorithmic traffic); together the ~650 tiles of a block
// are 2.5 rounds of workgroups, so most tiles run WHOLE (plain read-add-store into the gradient sink) and only a tail is cut
// into K slices (atomic adds) to level the last round.  Grid order = dispatch order: per problem its whole tiles first
// (`unsplit` of them, XCD-aware raster inside the problem), then per problem the slices of its remaining tiles.
struct GroupProblem {
    const bf16_t* A;       // dY [tokens][N]  (k-major "A": output rows = N)
    const bf16_t* B;       // X  [tokens][K]  (k-major "B": output cols = K)
    float* C;              // dW [N][K] fp32, accumulated into
    int64_t lda, ldb, ldc;
    int M, N, K;           // GEMM view: M = rows of dW, N = cols of dW, K = tokens
    int tiles_m, tiles_n, group_m;
    int unsplit, splits;   // tiles [0, unsplit) run whole; the others in `splits` K slices each
    int r1_start, r2_start;   // first block id of this problem in the whole-tile region / the sliced region
};
struct GroupParams {
    int nprob, r2_base;
    int overwrite;         // whole tiles STORE their result (dW = ...) instead of adding to it; the sliced tiles were zeroed first
    GroupProblem pr[KALLE_MAX_GROUP];
};

// overwrite mode: the 256 x 256 regions of the tiles that are cut into K slices (atomic adds) are cleared by this launch first
__global__ __launch_bounds__(256) void gemm3_group_zero_kernel(GroupParams gp) {
    int pi = 0, base = 0;
    const int bid = blockIdx.x >> 2, band = blockIdx.x & 3;   // four workgroups (64-row bands) per sliced tile, problems in order
    for (;;) {
        const int nb = gp.pr[pi].tiles_m * gp.pr[pi].tiles_n - gp.pr[pi].unsplit;
        if (bid < base + nb || pi + 1 >= gp.nprob) break;
        base += nb;
        ++pi;
    }
    const GroupProblem& q = gp.pr[pi];
    int tm, tn;
    gemm_tile_coords_plain(q.unsplit + bid - base, q.tiles_m, q.tiles_n, q.group_m, tm, tn);
    const int r0 = tm * 256 + band * 64, c0 = tn * 256;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = r0 + i / 64, c = c0 + (i & 63) * 4;
        if (r < q.M && c < q.N) *reinterpret_cast<f32x4*>(q.C + (int64_t)r * q.ldc + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

__global__ __launch_bounds__(512, 2) void gemm3_wgrad_group_kernel(GroupParams gp) {
    // blocks b, b + 8, ... share an XCD (and its L2): give every XCD a contiguous run of each region's work order, so that
    // the ~32 tiles it runs at a time are neighbours in a problem's grouped raster and share operand panels through L2
    const bool whole = (int)blockIdx.x < gp.r2_base;
    const int bid = whole ? xcd_contiguous(blockIdx.x, gp.r2_base)
                          : gp.r2_base + xcd_contiguous(blockIdx.x - gp.r2_base, gridDim.x - gp.r2_base);
    int pi = 0, tile, slice = 0;
    if (whole) {
        while (pi + 1 < gp.nprob && bid >= gp.pr[pi + 1].r1_start) ++pi;
        tile = bid - gp.pr[pi].r1_start;
    } else {
        while (pi + 1 < gp.nprob && bid >= gp.pr[pi + 1].r2_start) ++pi;
        const int local = bid - gp.pr[pi].r2_start, nb = gp.pr[pi].tiles_m * gp.pr[pi].tiles_n - gp.pr[pi].unsplit;
        tile = gp.pr[pi].unsplit + local % nb;
        slice = local / nb;
    }
    const GroupProblem& q = gp.pr[pi];
    GemmParams p{};
    p.A = q.A; p.B = q.B; p.C = q.C;
    p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
    p.M = q.M; p.N = q.N; p.K = q.K;
    p.alpha = 1.f;
    p.rows_per_batch = 1;
    p.accumulate = gp.overwrite ? 0 : 1;
    p.atomic = whole ? 0 : 1;
    const int ntiles = q.tiles_m * q.tiles_n, nk_all = (q.K + BK2 - 1) / BK2;
    // raster: the whole tiles and the sliced tiles are each a contiguous run of the problem's grouped tile order
    p.tiles_m = q.tiles_m; p.tiles_n = q.tiles_n; p.group_m = q.group_m;
    int kt0 = 0, nk = nk_all;
    if (!whole) {
        const int per = (nk_all + q.splits - 1) / q.splits;
        kt0 = slice * per;
        nk = min(per, nk_all - kt0);
        if (nk <= 0) return;
    }
    gemm3_tiles<true, true, true, 0, false, false>(p, tile, 0, tile + 1, ntiles, kt0, nk);
}

template <bool A_KM, bool B_KM, bool C_F32, int GLU = 0>
int launch3(const GemmParams& p, hipStream_t st) {
    constexpr int lds = GEMM3_LDS;
    static std::atomic<uint64_t> lds_ok{0};
    kalle_allow_lds(reinterpret_cast<const void*>(gemm3_kernel<A_KM, B_KM, C_F32, GLU>), lds, lds_ok);
    dim3 grid(p.tiles_m * p.tiles_n, p.splits), block(512);
    if (p.mix_na >= 0) {
        const int ntiles = p.tiles_m * p.tiles_n;
        grid = dim3(p.mix_na * p.mix_sa + (ntiles - p.mix_na) * (p.mix_sa + 1), 1);
    } else if (!A_KM) {
        // persistent: one workgroup per CU (160 KiB of LDS each) walks its tiles; KALLE_GEMM_PERSIST=0: one workgroup per tile
        // (the same code with a grid of `tiles` workgroups: every workgroup finds no next tile)
        if (p.splits != 1) return KALLE_ERR_UNSUPPORTED;
        static const bool persist = !(getenv("KALLE_GEMM_PERSIST") && atoi(getenv("KALLE_GEMM_PERSIST")) == 0);
        // KALLE_GEMM_GRID: workgroups of the persistent launch (experiment: fewer than one per CU)
        static const int grid_env = getenv("KALLE_GEMM_GRID") ? atoi(getenv("KALLE_GEMM_GRID")) & ~7 : 0;
        const int cus = grid_env >= 8 ? std::min(grid_env, kalle_cu_count()) : kalle_cu_count();
        if (persist && (int)grid.x > cus) {
            // experiment switch: "us" for every shape, or "glu2:us" for the fused SwiGLU backward only
            static const char* de = getenv("KALLE_GEMM_DEPHASE_US");
            GemmParams q = p;
            if (de && (int)grid.x >= 4 * cus) {
                const bool only_glu2 = !strncmp(de, "glu2:", 5);
                if (!only_glu2 || GLU == 2) q.dephase_ticks = (int)(atof(only_glu2 ? de + 5 : de) * 100.0);
            }
            grid.x = cus;
            // dynamic tile hand-out (KALLE_GEMM_DYNAMIC=0: static walk); a set is reused after SCHED_SETS persistent launches
            static const bool dynamic = !(getenv("KALLE_GEMM_DYNAMIC") && atoi(getenv("KALLE_GEMM_DYNAMIC")) == 0);
            static std::atomic<unsigned> next_set{0};
            // (a launch that is being captured into a HIP graph keeps the static walk: a replayed node would come back with the
            // same counter set while an eager launch on another stream may hold it)
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
            q.sched_set = dynamic && !capturing ? 1 + (int)(next_set.fetch_add(1, std::memory_order_relaxed) % SCHED_SETS) : 0;
            KALLE_LAUNCH((gemm3_kernel<A_KM, B_KM, C_F32, GLU>), grid, block, lds, st, q);
            return kalle_check_launch();
        }
    }
    KALLE_LAUNCH((gemm3_kernel<A_KM, B_KM, C_F32, GLU>), grid, block, lds, st, p);
    return kalle_check_launch();
}

int launch3_layout(const GemmParams& p, bool a_km, bool b_km, bool f32, hipStream_t st) {
    if (!a_km && !b_km) return f32 ? launch3<false, false, true>(p, st) : launch3<false, false, false>(p, st);
    if (!a_km && b_km) return f32 ? launch3<false, true, true>(p, st) : launch3<false, true, false>(p, st);
    if (a_km && b_km) return launch3<true, true, true>(p, st);
    return KALLE_ERR_UNSUPPORTED;
}

template <int WM, int WN, int TM>
int launch2_layout(const GemmParams& p, bool a_km, bool b_km, bool f32, hipStream_t st) {
    if (!a_km && !b_km) return f32 ? launch2<false, false, true, WM, WN, TM>(p, st) : launch2<false, false, false, WM, WN, TM>(p, st);
    if (!a_km && b_km) return f32 ? launch2<false, true, true, WM, WN, TM>(p, st) : launch2<false, true, false, WM, WN, TM>(p, st);
    if (a_km && b_km) return launch2<true, true, true, WM, WN, TM>(p, st);   // wgrad: fp32 out only
    return KALLE_ERR_UNSUPPORTED;
}

int g_force = -1;  // KALLE_GEMM=v1 | v2 (A/B testing); default: v2 where eligible
int force_mode() {
    if (g_force < 0) {
        const char* e = getenv("KALLE_GEMM");
        g_force = !e ? 0 : (!strcmp(e, "v1") ? 1 : (!strcmp(e, "v2") ? 2 : 0));
    }
    return g_force;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int kalle_gemm_v2_launch(GemmParams& p, bool a_km, bool b_km, bool f32, hipStream_t st) {
    p.mix_na = -1;
    p.mix_sa = 0;
    if (p.K & 7) return KALLE_ERR_UNSUPPORTED;        // (a ragged last K-tile is fine: K % 64 in multiples of 8)
    if (a_km && !b_km) return KALLE_ERR_UNSUPPORTED;
    if (a_km && !f32) return KALLE_ERR_UNSUPPORTED;
    static const int min_m = getenv("KALLE_V2_MIN_M") ? atoi(getenv("KALLE_V2_MIN_M")) : 256;
    if (p.M < min_m || p.N < 128) return KALLE_ERR_UNSUPPORTED;
    if (a_km && (p.M & 7)) return KALLE_ERR_UNSUPPORTED;
    const int nk = (p.K + BK2 - 1) / BK2;
    if (p.glu_mode) {
        // fused SwiGLU: 256 x 256 kernel only; forward pairs x/gate weight rows inside each wave's 64 tile columns
        if (a_km || f32 || p.gate || p.residual || p.row_mask || p.c_rpb || p.accumulate) return KALLE_ERR_UNSUPPORTED;
        if (p.glu_inner % 128 || p.N != (p.glu_mode == 1 ? 2 * p.glu_inner : p.glu_inner)) return KALLE_ERR_UNSUPPORTED;
        if (p.glu_mode == 1 && b_km) return KALLE_ERR_UNSUPPORTED;
        if (p.glu_mode == 2 && (!b_km || p.bias)) return KALLE_ERR_UNSUPPORTED;
        p.tiles_m = (p.M + 255) / 256;
        p.tiles_n = p.glu_mode == 1 ? p.glu_inner / 128 : (p.glu_inner + 255) / 256;
        p.tile_n = 256;
        p.group_m = p.tiles_m < 4 ? p.tiles_m : 4;
        p.splits = 1;
        p.atomic = 0;
        p.ktiles_per_split = nk;
        if (p.glu_mode == 1) return launch3<false, false, false, 1>(p, st);
        return launch3<false, true, false, 2>(p, st);
    }
    const bool plain = !p.bias && !p.gate && !p.residual && !p.row_mask && p.c_rpb == 0;
    const bool can_split = f32 && plain && a_km && nk >= 16;
    static int tile_env = -1;
    if (tile_env < 0) {
        const char* e = getenv("KALLE_GEMM_TILE");
        tile_env = e ? atoi(e) : 0;
    }
    // modelled time of a configuration: MFMA work / (tile rate x wave-quantisation efficiency) + split-K atomic bytes
    const double flops = 2.0 * p.M * p.N * p.K;
    double best = 1e30;
    int best_bn = 128, best_s = 1;
    for (int bn = 128; bn <= 256; bn += 128) {
        if (tile_env && bn != tile_env) continue;
        if (bn == 256 && p.N < 256) continue;
        const double rate = bn == 256 ? 1150e12 : 1000e12;
        const int tiles = ((p.M + 255) / 256) * ((p.N + bn - 1) / bn);
        for (int s = 1; s <= (can_split ? 16 : 1) && (s == 1 || nk / s >= 8); ++s) {
            const int blocks = tiles * s;
            const double eff = (double)blocks / (((blocks + 255) / 256) * 256.0);
            const double t = flops / (rate * eff) + (s > 1 ? (double)s * p.M * p.N * 4.0 / 2.0e12 : 0.0);
            if (t < best * 0.98) { best = t; best_bn = bn; best_s = s; }
        }
    }
    // Weight gradients on the 256 x 256 kernel: mixed split-K.  `na` tiles get `sa` K slices, the others sa + 1, the longer
    // slices are dispatched first: the last round of workgroups then consists of short slices instead of leaving most CUs idle
    // (288 tiles x 3 slices = 3.4 rounds -> 4 with uniform splitting).  The plan comes from replaying the dispatch on 256 CUs
    // (time of a slice = 10 us + 1.6 us per K-tile, atomics at 2 TB/s - fitted to tools/wgrad_sweep.py) and is cached per shape.
    static const bool nomix_env = getenv("KALLE_GEMM_NOMIX") != nullptr, debug_env = getenv("KALLE_GEMM_DEBUG") != nullptr;
    static const char* const mix_env = getenv("KALLE_GEMM_MIX");          // experiment switches: read once per process
    if (can_split && best_bn == 256 && !tile_env && !nomix_env) {
        struct Plan { int M, N, kb, sa, na; };          // kb: K-tiles / 16 (a plan is valid for any K; nearby K share it)
        static thread_local Plan cache[32];
        static thread_local int ncache = 0, victim = 0;
        const Plan* hit = nullptr;
        int sa_sel = 0, na_sel = -1;
        for (int i = 0; i < ncache; ++i)
            if (cache[i].M == p.M && cache[i].N == p.N && cache[i].kb == nk / 16) hit = &cache[i];
        if (!hit) {
            const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
            auto replay = [&](int sa, int na) {        // makespan (us) of the dispatch order on 256 CUs + atomic traffic
                double cu[256];                         // min-heap of the times at which the CUs fall free
                for (int c = 0; c < 256; ++c) cu[c] = 0.0;
                auto run = [&](int tiles, int slices) {
                    if (tiles <= 0 || slices <= 0) return;
                    const int per = (nk + slices - 1) / slices;
                    for (int sl = 0; sl < slices; ++sl) {
                        const int k = nk - sl * per < per ? nk - sl * per : per;
                        if (k <= 0) break;
                        const double d = 10.0 + 1.6 * k;
                        for (int t = 0; t < tiles; ++t) {
                            std::pop_heap(cu, cu + 256, std::greater<double>());
                            cu[255] += d;
                            std::push_heap(cu, cu + 256, std::greater<double>());
                        }
                    }
                };
                run(na, sa);
                run(ntiles - na, sa + 1);
                double mk = 0.0;
                for (int c = 0; c < 256; ++c) mk = cu[c] > mk ? cu[c] : mk;
                const double savg = ((double)na * sa + (double)(ntiles - na) * (sa + 1)) / ntiles;
                return mk + (savg > 1.0 ? savg * p.M * p.N * 4.0 / 2.0e12 * 1e6 : 0.0);
            };
            Plan best_plan{p.M, p.N, nk / 16, 0, -1};
            double t_uniform = 1e30, t_best = 1e30;
            const int step = ntiles / 12 > 4 ? (ntiles / 12 + 3) & ~3 : 4;
            // uniform plans: every slice count; mixed plans: around the uniform model's choice (best_s slices)
            for (int sl = 1; sl <= 16 && (sl == 1 || nk / sl >= 8); ++sl) {
                const double t = replay(sl - 1, 0);
                if (t < t_uniform) t_uniform = t;
            }
            for (int sa = best_s > 3 ? best_s - 3 : 1; sa <= best_s + 1 && sa <= 15; ++sa) {
                if (nk / (sa + 1) < 8) break;
                for (int na = step; na < ntiles; na += step) {
                    const double t = replay(sa, na);
                    if (t < t_best) { t_best = t; best_plan.sa = sa; best_plan.na = na; }
                }
            }
            if (!(best_plan.na > 0 && t_best < 0.97 * t_uniform)) best_plan.na = -1;   // not worth leaving the uniform plan
            if (ncache < 32) cache[ncache++] = best_plan;
            else { cache[victim] = best_plan; victim = (victim + 1) & 31; }
            sa_sel = best_plan.sa;
            na_sel = best_plan.na;
        } else {
            sa_sel = hit->sa;
            na_sel = hit->na;
        }
        if (na_sel > 0 && nk / (sa_sel + 1) < 8) na_sel = -1;    // (a cached plan of a longer K)
        if (debug_env) fprintf(stderr, "[kalle gemm] %d x %d x %d: mixed split sa=%d na=%d (hit=%d)\n", p.M, p.N, p.K, sa_sel, na_sel, hit != nullptr);
        if (na_sel > 0) { p.mix_na = na_sel; p.mix_sa = sa_sel; best_s = sa_sel + 1; }
    }
    if (const char* e = mix_env) {      // "sa,na": experiment override
        int sa = 0, na = 0;
        if (can_split && sscanf(e, "%d,%d", &sa, &na) == 2 && sa >= 1 && best_bn == 256 && nk / (sa + 1) >= 8) {
            const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
            p.mix_sa = sa;
            p.mix_na = na < ntiles ? na : ntiles;
            best_s = sa + 1;                             // (>1: atomic epilogue + cleared C below)
        }
    }
    static const int s_env = getenv("KALLE_GEMM_SPLITS") ? atoi(getenv("KALLE_GEMM_SPLITS")) : 0;
    if (s_env > 0 && can_split && nk / s_env >= 8) best_s = s_env;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + best_bn - 1) / best_bn;
    p.tile_n = best_bn;
    static int gm_env = -1;
    if (gm_env < 0) { const char* e = getenv("KALLE_GEMM_GM"); gm_env = e ? atoi(e) : 0; }
    const int gm_want = gm_env > 0 ? gm_env : 4;
    p.group_m = p.tiles_m < gm_want ? p.tiles_m : gm_want;
    p.splits = best_s;
    p.atomic = best_s > 1;
    if (p.atomic && !p.accumulate) {
        if (hipMemset2DAsync(p.C, p.ldc * sizeof(float), 0, p.N * sizeof(float), p.M, st) != hipSuccess)
            return KALLE_ERR_LAUNCH;
    }
    p.ktiles_per_split = (nk + p.splits - 1) / p.splits;
    p.splits = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
    if (best_bn == 256) return launch3_layout(p, a_km, b_km, f32, st);
    return launch2_layout<4, 2, 4>(p, a_km, b_km, f32, st);
}

// ---- few rows (M <= 4096): split-K into a caller-provided fp32 scratch + finishing pass -----------------------------------
// A GEMM whose output has only a few 256-row tile rows cannot fill 256 CUs with output tiles: the DiT forward at generation
// batch sizes (B = 1 with CFG: 252 rows) gives 12 - 96 tiles, a B = 16 train step 96 - 384.  K is cut into slices that add
// their partial tiles as plain fp32 slabs [slice][M][N] into the caller's scratch; the finishing pass sums the slabs in slice order
// (bitwise reproducible) and applies the whole epilogue (alpha, bias, adaLN gate, row mask, residual, accumulate, output-row
// remap, fused SwiGLU forward) while writing C.
namespace {
template <bool C_F32>
__global__ __launch_bounds__(256) void gemm_finish_kernel(GemmParams p, const float* __restrict__ ws, int nslab) {
    const int64_t slab = (int64_t)p.M * p.N;
    auto ld4 = [&](int64_t off) {               // sum of the K slices' partial results, always in slice order
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + off);
        for (int s = 1; s < nslab; ++s) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(ws + s * slab + off);
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        return a;
    };
    const int cols4 = (p.glu_mode == 1 ? p.glu_inner : p.N) >> 2;
    const int64_t total = (int64_t)p.M * cols4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int gm = (int)(i / cols4), gn = (int)(i - (int64_t)gm * cols4) * 4;
        if (p.glu_mode == 1) {
            // fused SwiGLU forward (transformer.py:216-219): h = x W^T + b [M][2 inner] (bf16), act = h_x * silu(h_gate)
            const f32x4 xv = ld4((int64_t)gm * p.N + gn);
            const f32x4 gv = ld4((int64_t)gm * p.N + p.glu_inner + gn);
            float x[4] = {xv[0], xv[1], xv[2], xv[3]}, g[4] = {gv[0], gv[1], gv[2], gv[3]};
            if (p.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { x[j] += p.bias[gn + j]; g[j] += p.bias[p.glu_inner + gn + j]; }
            }
            i32x2 hx, hg, av;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                hx[e] = (int)pack_bf16x2(x[2 * e], x[2 * e + 1]);
                hg[e] = (int)pack_bf16x2(g[2 * e], g[2 * e + 1]);
                const float x0 = bf16lo((uint32_t)hx[e]), x1 = bf16hi((uint32_t)hx[e]);
                const float g0 = bf16lo((uint32_t)hg[e]), g1 = bf16hi((uint32_t)hg[e]);
                av[e] = (int)pack_bf16x2(x0 * siluf_(g0), x1 * siluf_(g1));
            }
            bf16_t* hp = reinterpret_cast<bf16_t*>(p.C) + (int64_t)gm * p.ldc;
            *reinterpret_cast<i32x2*>(hp + gn) = hx;
            *reinterpret_cast<i32x2*>(hp + p.glu_inner + gn) = hg;
            *reinterpret_cast<i32x2*>(static_cast<bf16_t*>(p.glu_aux) + (int64_t)gm * p.glu_inner + gn) = av;
            continue;
        }
        const f32x4 a = ld4((int64_t)gm * p.N + gn);
        float v[4] = {a[0], a[1], a[2], a[3]};
        const int64_t crow = gemm_crow(p, gm);
        gemm_epilogue4(p, gm, gn, crow, v, true);
        if constexpr (C_F32) {
            float* cp = reinterpret_cast<float*>(p.C) + crow * p.ldc + gn;
            if (p.accumulate) {
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += c0[j];
            }
            *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            i32x2 o;
            o[0] = (int)pack_bf16x2(v[0], v[1]);
            o[1] = (int)pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<i32x2*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gn) = o;
        }
    }
}
}  // namespace

// ---- few rows, k-contiguous operands (every nn.Linear of the sampling path: M = 252 for one clip with CFG) -------------------
// One workgroup's K loop is bound by the LDS-DMA ingest of its CU (~60 GB/s: 0.8 us per 64-deep K-tile of a 256 x 128 tile), and
// every fp32 byte a K slice parks in a slab is paid twice (1.7 TB/s effective at this size: the write-back at the kernel's end, then
// the finishing pass) - so instead of cutting K, the OUTPUT is cut into small tiles (64 x 64 with one wave, 128 x 64, 128 x 128) until
// about a workgroup per CU exists, each over the whole K with the full epilogue in the same launch; only K > 2048 is also cut
// into slices (slabs + finishing pass).  `cfg` packs the choice for kalle_gemm_last_plan: 5 | WM << 8 | WN << 12 | splits << 16.
template <bool C_F32, int GLU>
int launch_skinny_tile(int wm, int wn, const GemmParams& q, hipStream_t st, bool b_km = false) {
    static const bool one_group = getenv("KALLE_SKINNY_KS") && atoi(getenv("KALLE_SKINNY_KS")) == 1;    // experiment switch
    if (b_km) {             // k-major weights (data gradients): 128-column tiles only
        if constexpr (GLU == 0) {
            if (wm == 2 && wn == 2) return launch2_ks2<true, C_F32, 2, 2, 0, 4>(q, st);
        }
        return KALLE_ERR_UNSUPPORTED;
    }
    if (!one_group) {       // two wave groups per tile (one per 32-deep k-step): two waves per SIMD cover each other
        if (wm == 1 && wn == 1) return launch2_ks2<false, C_F32, 1, 1, GLU, 5>(q, st);         // 80 KiB
        if (wm == 2 && wn == 1) return launch2_ks2<false, C_F32, 2, 1, GLU, 5>(q, st);         // 120 KiB
        if (wm == 2 && wn == 2) return launch2_ks2<false, C_F32, 2, 2, GLU, 4>(q, st);         // 128 KiB
    }
    if (wm == 1 && wn == 1) return launch2<false, false, C_F32, 1, 1, 4, GLU, 5>(q, st);
    if (wm == 2 && wn == 1) return launch2<false, false, C_F32, 2, 1, 4, GLU, 5>(q, st);
    if (wm == 2 && wn == 2) return launch2<false, false, C_F32, 2, 2, 4, GLU, 4>(q, st);
    return KALLE_ERR_UNSUPPORTED;
}

int kalle_gemm_skinny_launch(const GemmParams& pin, bool a_km, bool b_km, bool f32, void* ws, int64_t ws_bytes, hipStream_t st,
                             int* cfg) {
    static const char* env = getenv("KALLE_SKINNY");           // "0": off; "wm,wn,splits": forced configuration (experiments)
    if (env && env[0] == '0' && !env[1]) return KALLE_ERR_UNSUPPORTED;
    static const int max_m = getenv("KALLE_SKINNY_MAX_M") ? atoi(getenv("KALLE_SKINNY_MAX_M")) : 2048;
    if (a_km || pin.M > max_m || (pin.K & 7) || (pin.N & 63) || pin.atomic) return KALLE_ERR_UNSUPPORTED;
    if (b_km && ((pin.N & 127) || pin.glu_mode)) return KALLE_ERR_UNSUPPORTED;
    if (pin.glu_mode == 2 || (pin.glu_mode == 1 && (f32 || pin.N != 2 * pin.glu_inner || (pin.glu_inner & 31) || pin.gate ||
                                                    pin.residual || pin.row_mask || pin.c_rpb || pin.accumulate)))
        return KALLE_ERR_UNSUPPORTED;
    // outputs with a chip's worth of 256 x 256 tiles belong to the big-tile kernels (twice the flops per ingested byte)
    if ((int64_t)((pin.M + 255) / 256) * ((pin.N + 255) / 256) >= 256) return KALLE_ERR_UNSUPPORTED;
    const int nk = (pin.K + BK2 - 1) / BK2;
    const int ncol = pin.glu_mode == 1 ? pin.glu_inner * 2 : pin.N;
    // candidate tiles, smallest first; cost (us) = K-tiles per workgroup x the ingest time of the tiles sharing a CU + slab traffic
    static const int cand[3][2] = {{1, 1}, {2, 1}, {2, 2}};
    int wm = 0, wn = 0, splits = 1;
    double best = 1e30;
    int fwm = 0, fwn = 0, fs = 0;
    const bool forced = env && sscanf(env, "%d,%d,%d", &fwm, &fwn, &fs) == 3;
    for (int c = 0; c < 3; ++c) {
        const int bm = cand[c][0] * 64, bn = cand[c][1] * 64;
        if (pin.glu_mode == 1 && (pin.glu_inner % (bn / 2))) continue;
        if (b_km && bn < 128) continue;
        const int tiles = ((pin.M + bm - 1) / bm) * ((ncol + bn - 1) / bn);
        for (int sp = 1; sp <= 8; ++sp) {
            if (sp > 1 && (nk / sp < 12 || !ws || pin.glu_mode)) break;
            if (sp > 1 && (int64_t)sp * pin.M * pin.N * 4 > ws_bytes) break;
            // more than a few tile rows (training at small batch): measured against the 256-row kernels, the small tiles win only
            // while they fit one round of workgroups and K is moderate (B = 16: 2016 x 1536 x 1536 30 -> 25 us, x 4608 64 -> 50,
            // x 6144 78 -> 65; but 2016 x 4608 x 1536 in 2.25 rounds 57 -> 65 and K = 12288 115 -> 122)
            if (pin.M > 512 && (tiles * sp > 256 || nk > 128)) continue;
            const double wgs = (double)tiles * sp;
            // 64 x 64 tiles (80 KiB of LDS) fit two to a CU and share its ingest - a fractional round is the right price; the larger
            // tiles (120 / 128 KiB) run one per CU, so 288 of them ARE two rounds (M = 504, N = 4608: 128 x 64 tiles 28.3 us where
            // 144 tiles of 128 x 128 take 19)
            const double rounds = c == 0 ? (wgs / 256.0 > 1.0 ? wgs / 256.0 : 1.0) : (double)(((int64_t)wgs + 255) / 256);
            const double tk = (bm + bn) * 128.0 / 60e3;                            // us per K-tile: the CU's LDS-DMA ingest ...
            const double tk_floor = 0.30;                                          // ... or barrier + fragment reads + 16 MFMAs per wave
            double t = 2.5 + rounds * (3.0 + (double)((nk + sp - 1) / sp) * (tk > tk_floor ? tk : tk_floor));
            // slabs: second launch + write + read back.  (The slabs of these shapes are a few MB that the finishing pass finds in L2 /
            // Infinity Cache: 3 TB/s fits M = 252 ... 1008 x 1536 x 6144, where 1.7 TB/s kept M = 504 on whole-K 64 x 64 tiles at 37 us)
            if (sp > 1) t += 4.0 + 2.0 * sp * pin.M * (double)pin.N * 4.0 / 3.0e6;
            const bool pick = forced ? (cand[c][0] == fwm && cand[c][1] == fwn && sp == fs) : t < best;
            if (pick) { best = t; wm = cand[c][0]; wn = cand[c][1]; splits = sp; }
        }
    }
    if (!wm) return KALLE_ERR_UNSUPPORTED;
    const int bm = wm * 64, bn = wn * 64;
    GemmParams q = pin;
    q.tiles_m = (pin.M + bm - 1) / bm;
    q.tiles_n = (ncol + bn - 1) / bn;
    q.tile_n = bn;
    q.group_m = q.tiles_m < 4 ? q.tiles_m : 4;
    q.atomic = 0;
    q.mix_na = -1;
    q.ktiles_per_split = (nk + splits - 1) / splits;
    q.splits = (nk + q.ktiles_per_split - 1) / q.ktiles_per_split;
    q.slab_stride = 0;
    if (cfg) *cfg = 5 | (wm << 8) | (wn << 12) | (q.splits << 16);
    if (q.splits == 1) {
        if (pin.glu_mode == 1) return launch_skinny_tile<false, 1>(wm, wn, q, st);
        return f32 ? launch_skinny_tile<true, 0>(wm, wn, q, st, b_km) : launch_skinny_tile<false, 0>(wm, wn, q, st, b_km);
    }
    // K slices into fp32 slabs (plain epilogue-free stores), then the finishing pass with the caller's epilogue
    GemmParams sl{};
    sl.A = pin.A; sl.B = pin.B; sl.C = ws;
    sl.lda = pin.lda; sl.ldb = pin.ldb; sl.ldc = pin.N;
    sl.M = pin.M; sl.N = pin.N; sl.K = pin.K;
    sl.alpha = 1.f;
    sl.rows_per_batch = 1;
    sl.tiles_m = q.tiles_m; sl.tiles_n = q.tiles_n; sl.tile_n = bn; sl.group_m = q.group_m;
    sl.mix_na = -1;
    sl.ktiles_per_split = q.ktiles_per_split; sl.splits = q.splits;
    sl.slab_stride = (int64_t)pin.M * pin.N;
    const int rc = launch_skinny_tile<true, 0>(wm, wn, sl, st, b_km);
    if (rc != KALLE_OK) return rc;
    const int64_t work = (int64_t)pin.M * (pin.N >> 2);
    const int grid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    if (f32) KALLE_LAUNCH(gemm_finish_kernel<true>, dim3(grid), dim3(256), 0, st, pin, static_cast<const float*>(ws), q.splits);
    else KALLE_LAUNCH(gemm_finish_kernel<false>, dim3(grid), dim3(256), 0, st, pin, static_cast<const float*>(ws), q.splits);
    return kalle_check_launch();
}

// returns KALLE_ERR_UNSUPPORTED when the shape is better served by the ordinary path (the caller goes on to it)
int kalle_gemm_few_rows_launch(const GemmParams& pin, bool a_km, bool b_km, bool f32, void* ws, int64_t ws_bytes, hipStream_t st) {
    if (a_km || !ws || pin.M > 4096 || (pin.K & 7) || (pin.N & 7)) return KALLE_ERR_UNSUPPORTED;
    if (pin.glu_mode == 2 || (pin.glu_mode == 1 && (b_km || f32 || pin.N != 2 * pin.glu_inner || (pin.glu_inner & 3))))
        return KALLE_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(ws) & 15) return KALLE_ERR_UNSUPPORTED;
    const int nk = (pin.K + BK2 - 1) / BK2;
    // (256 x 64 tiles with 4 waves - twice the column tiles, half the K slices - measured 5 % slower on the sampling step)
    constexpr int bn = 128;
    const int tiles = ((pin.M + 255) / 256) * ((pin.N + bn - 1) / bn);
    if (tiles >= 192) return KALLE_ERR_UNSUPPORTED;                 // enough output tiles on their own
    static const int target = getenv("KALLE_FEW_ROWS_TARGET") ? atoi(getenv("KALLE_FEW_ROWS_TARGET")) : 320;
    int splits = (target + tiles - 1) / tiles;                       // ~1.25 workgroups per CU
    // at least two K-tiles per slice; with more than a couple of tile rows (training at small batch) a slice must be long
    // enough (16 K-tiles) to pay for its slab: M x N x 4 bytes written and read back per slice
    const int min_per = pin.M > 512 ? 16 : 2;
    if (splits > nk / min_per) splits = nk / min_per;
    const int64_t slab = (int64_t)pin.M * pin.N;
    if (splits > ws_bytes / (4 * slab)) splits = (int)(ws_bytes / (4 * slab));   // one fp32 [M][N] slab per slice
    if (splits < 2) return KALLE_ERR_UNSUPPORTED;
    GemmParams q{};
    q.A = pin.A; q.B = pin.B; q.C = ws;
    q.lda = pin.lda; q.ldb = pin.ldb; q.ldc = pin.N;
    q.M = pin.M; q.N = pin.N; q.K = pin.K;
    q.alpha = 1.f;
    q.rows_per_batch = 1;
    q.tiles_m = (pin.M + 255) / 256; q.tiles_n = (pin.N + bn - 1) / bn; q.tile_n = bn;
    q.group_m = q.tiles_m < 4 ? q.tiles_m : 4;
    q.atomic = 0;
    q.mix_na = -1;
    q.ktiles_per_split = (nk + splits - 1) / splits;
    q.splits = (nk + q.ktiles_per_split - 1) / q.ktiles_per_split;
    q.slab_stride = slab;
    const int rc = b_km ? launch2<false, true, true, 4, 2, 4>(q, st) : launch2<false, false, true, 4, 2, 4>(q, st);
    if (rc != KALLE_OK) return rc;
    GemmParams f = pin;
    const int64_t work = (int64_t)pin.M * ((pin.glu_mode == 1 ? pin.glu_inner : pin.N) >> 2);
    const int grid = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    if (f32) KALLE_LAUNCH(gemm_finish_kernel<true>, dim3(grid), dim3(256), 0, st, f, static_cast<const float*>(ws), q.splits);
    else KALLE_LAUNCH(gemm_finish_kernel<false>, dim3(grid), dim3(256), 0, st, f, static_cast<const float*>(ws), q.splits);
    return kalle_check_launch();
}

static unsigned long long* g_stamps = nullptr;      // diagnostics only (kalle_gemm_debug_stamps)
extern "C" int kalle_gemm_debug_stamps(void* buf) { g_stamps = static_cast<unsigned long long*>(buf); return KALLE_OK; }

namespace {
__global__ __launch_bounds__(256) void hold_cus_kernel(int ticks) {
    extern __shared__ char hold_smem[];
    if (ticks < 0) hold_smem[threadIdx.x] = 1;          // (keeps the allocation alive)
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime() + (unsigned long long)ticks;
    while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(64);
}
}  // namespace
extern "C" int kalle_debug_hold_cus(int nwg, int lds_bytes, int microseconds, void* stream) {
    if (nwg <= 0 || nwg > 4096 || lds_bytes < 0 || lds_bytes > 65536 || microseconds < 0 || microseconds > 1000000) return KALLE_ERR_ARG;
    KALLE_LAUNCH(hold_cus_kernel, dim3(nwg), dim3(256), lds_bytes, static_cast<hipStream_t>(stream), microseconds * 100);
    return kalle_check_launch();
}

static thread_local int g_last_plan = 0;
extern "C" int kalle_gemm_last_plan(void) { return g_last_plan; }

extern "C" int kalle_gemm_bf16(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                               void* C, int64_t ldc, int c_dtype, int M, int N, int K,
                               const kalle_gemm_epilogue* ep, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return KALLE_ERR_ARG;
    if ((N & 7) || (lda & 7) || (ldb & 7) || (ldc & 7)) return KALLE_ERR_ARG;
    if (!a_kmajor && (K & 7)) return KALLE_ERR_ARG;
    if (!b_kmajor && (K & 7)) return KALLE_ERR_ARG;
    if (a_kmajor && (M & 7)) return KALLE_ERR_ARG;
    if (!al16(A) || !al16(B) || !al16(C)) return KALLE_ERR_ARG;
    if (c_dtype != KALLE_BF16 && c_dtype != KALLE_F32) return KALLE_ERR_ARG;
    GemmParams p{};
    p.A = static_cast<const bf16_t*>(A);
    p.B = static_cast<const bf16_t*>(B);
    p.C = C;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.alpha = 1.f;
    p.rows_per_batch = 1;
    if (ep) {
        p.bias = ep->bias;
        p.gate = ep->gate; p.ldg = ep->ldg; p.rows_per_batch = ep->rows_per_batch > 0 ? ep->rows_per_batch : 1;
        p.residual = ep->residual; p.ldr = ep->ldr;
        p.accumulate = ep->accumulate;
        if (ep->alpha != 0.f) p.alpha = ep->alpha;
        p.row_mask = ep->row_mask;
        p.c_rpb = ep->c_rows_per_batch; p.c_brows = ep->c_batch_rows; p.c_roff = ep->c_row_offset;
        p.glu_mode = ep->glu_mode; p.glu_inner = ep->glu_inner; p.glu_aux = ep->glu_aux; p.glu_dbias = ep->glu_dbias;
        if (p.glu_mode && (!p.glu_aux || p.glu_inner <= 0)) return KALLE_ERR_ARG;
        if (p.accumulate && c_dtype != KALLE_F32) return KALLE_ERR_ARG;
        if ((p.bias && !al16(p.bias)) || (p.gate && (!al16(p.gate) || (p.ldg & 3))) ||
            (p.residual && (!al16(p.residual) || (p.ldr & 3))))
            return KALLE_ERR_ARG;
    }
    p.stamps = g_stamps;
    static const int dbg_env = getenv("KALLE_GEMM_DBG") ? atoi(getenv("KALLE_GEMM_DBG")) : 0;   // knock-out timing (diagnostics)
    p.dbg = dbg_env;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool f32 = c_dtype == KALLE_F32;
    if (force_mode() != 1) {
        int cfg = 0;
        const int rc = kalle_gemm_skinny_launch(p, a_kmajor != 0, b_kmajor != 0, f32, ep ? ep->workspace : nullptr,
                                                ep ? ep->workspace_bytes : 0, st, &cfg);
        if (rc != KALLE_ERR_UNSUPPORTED) {
            g_last_plan = cfg;
            return rc;
        }
    }
    if (ep && ep->workspace && force_mode() != 1) {
        const int rc = kalle_gemm_few_rows_launch(p, a_kmajor != 0, b_kmajor != 0, f32, ep->workspace, ep->workspace_bytes, st);
        if (rc != KALLE_ERR_UNSUPPORTED) {
            g_last_plan = 4 | (1 << 8);
            return rc;
        }
    }
    if (force_mode() != 1) {
        const int rc = kalle_gemm_v2_launch(p, a_kmajor != 0, b_kmajor != 0, f32, st);
        if (rc != KALLE_ERR_UNSUPPORTED) {
            g_last_plan = (p.tile_n == 256 ? 3 : 2) | (p.splits << 8);
            return rc;
        }
        static const bool strict_env = getenv("KALLE_GEMM_STRICT") != nullptr;
        if (force_mode() == 2 && strict_env) return rc;
    }
    if (p.glu_mode) return KALLE_ERR_UNSUPPORTED;   // fused SwiGLU exists only in the 256x256 kernel: caller un-fuses
    p.tiles_n = (N + 127) / 128;
    g_last_plan = 1;
    return kalle_gemm_v1_launch(p, a_kmajor != 0, b_kmajor != 0, f32, st);
}

// ---- kalle_gemm_wgrad_group: host side -----------------------------------------------------------------------------------
namespace {
struct GroupPlan { uint64_t key; int ntot; int unsplit_total, splits; };

// makespan (us) of a plan on 256 CUs: whole tiles first, then the slices; a slice costs 10 us + 1.6 us per K-tile (fitted to
// tools/wgrad_sweep.py), atomics at 2 TB/s for the sliced tiles only
double replay_group(const GroupProblem* pr, int n, int unsplit_total, int splits) {
    double cu[256];
    for (int c = 0; c < 256; ++c) cu[c] = 0.0;
    auto put = [&](double d) {
        std::pop_heap(cu, cu + 256, std::greater<double>());
        cu[255] += d;
        std::push_heap(cu, cu + 256, std::greater<double>());
    };
    int left = unsplit_total;
    double atomic_bytes = 0.0;
    int rest[KALLE_MAX_GROUP];
    for (int i = 0; i < n; ++i) {
        const int t = pr[i].tiles_m * pr[i].tiles_n, u = left < t ? left : t;
        left -= u;
        rest[i] = t - u;
        const double d = 10.0 + 1.6 * ((pr[i].K + BK2 - 1) / BK2);
        for (int k = 0; k < u; ++k) put(d);
    }
    for (int i = 0; i < n; ++i) {
        if (!rest[i]) continue;
        const int nk = (pr[i].K + BK2 - 1) / BK2, per = (nk + splits - 1) / splits;
        for (int sl = 0; sl < splits; ++sl) {
            const int k = nk - sl * per < per ? nk - sl * per : per;
            if (k <= 0) break;
            for (int t = 0; t < rest[i]; ++t) put(10.0 + 1.6 * k);
        }
        if (splits > 1) atomic_bytes += (double)rest[i] * splits * 256.0 * 256.0 * 4.0;
    }
    double mk = 0.0;
    for (int c = 0; c < 256; ++c) mk = cu[c] > mk ? cu[c] : mk;
    return mk + atomic_bytes / 2.0e12 * 1e6;
}
}  // namespace

extern "C" int kalle_gemm_wgrad_group(const kalle_wgrad_problem* problems, int nprob, int overwrite, void* stream) {
    if (!problems || nprob <= 0 || nprob > KALLE_MAX_GROUP) return KALLE_ERR_ARG;
    GroupParams gp{};
    gp.nprob = nprob;
    gp.overwrite = overwrite ? 1 : 0;
    int ntot = 0, min_nk = 1 << 30;
    uint64_t key = 1469598103934665603ull;
    for (int i = 0; i < nprob; ++i) {
        const kalle_wgrad_problem& w = problems[i];
        if (!w.dy || !w.x || !w.dw || w.N <= 0 || w.K <= 0 || w.tokens <= 0) return KALLE_ERR_ARG;
        if ((w.N & 7) || (w.K & 7) || (w.lddy & 7) || (w.ldx & 7) || (w.lddw & 3)) return KALLE_ERR_ARG;
        if (!al16(w.dy) || !al16(w.x) || !al16(w.dw)) return KALLE_ERR_ARG;
        if (w.tokens & 7) return KALLE_ERR_UNSUPPORTED;       // (the caller falls back to one kalle_gemm_bf16 per gradient)
        GroupProblem& q = gp.pr[i];
        q.A = static_cast<const bf16_t*>(w.dy); q.B = static_cast<const bf16_t*>(w.x); q.C = w.dw;
        q.lda = w.lddy; q.ldb = w.ldx; q.ldc = w.lddw;
        q.M = w.N; q.N = w.K; q.K = w.tokens;
        q.tiles_m = (q.M + 255) / 256; q.tiles_n = (q.N + 255) / 256;
        q.group_m = q.tiles_m < 4 ? q.tiles_m : 4;
        ntot += q.tiles_m * q.tiles_n;
        min_nk = (q.K + BK2 - 1) / BK2 < min_nk ? (q.K + BK2 - 1) / BK2 : min_nk;
        for (uint64_t v : {(uint64_t)w.N, (uint64_t)w.K, (uint64_t)(w.tokens / 1024)}) key = (key ^ v) * 1099511628211ull;
    }
    // plan: how many tiles run whole (a multiple of the 256 CUs, or all) and into how many slices the others are cut
    static thread_local GroupPlan cache[16];
    static thread_local int ncache = 0, victim = 0;
    const GroupPlan* hit = nullptr;
    for (int i = 0; i < ncache; ++i)
        if (cache[i].key == key && cache[i].ntot == ntot) hit = &cache[i];
    GroupPlan plan{key, ntot, ntot, 1};
    static const char* env_plan = getenv("KALLE_WGRAD_GROUP_PLAN");      // "unsplit,splits": experiment override
    if (env_plan && sscanf(env_plan, "%d,%d", &plan.unsplit_total, &plan.splits) == 2) {
        plan.unsplit_total = plan.unsplit_total < ntot ? (plan.unsplit_total < 0 ? 0 : plan.unsplit_total) : ntot;
        plan.splits = plan.splits < 1 ? 1 : plan.splits;
    } else if (hit) {
        plan = *hit;
    } else {
        double best = 1e30;
        for (int u = 0;; u += 256) {
            const int ut = u < ntot ? u : ntot;
            // (one slice per tile IS a whole tile, only through the atomic path: sp = 1 belongs to ut == ntot alone)
            for (int sp = ut == ntot ? 1 : 2; sp <= 12 && (sp == 1 || min_nk / sp >= 8); ++sp) {
                const double t = replay_group(gp.pr, nprob, ut, sp);
                if (t < best * 0.995) { best = t; plan.unsplit_total = ut; plan.splits = sp; }
            }
            if (ut == ntot) break;
        }
        if (ncache < 16) cache[ncache++] = plan;
        else { cache[victim] = plan; victim = (victim + 1) & 15; }
    }
    if (min_nk / plan.splits < 1) plan.splits = 1;
    if (plan.splits == 1) plan.unsplit_total = ntot;
    int left = plan.unsplit_total, b1 = 0;
    for (int i = 0; i < nprob; ++i) {
        GroupProblem& q = gp.pr[i];
        const int t = q.tiles_m * q.tiles_n;
        q.unsplit = left < t ? left : t;
        left -= q.unsplit;
        q.splits = plan.splits;
        q.r1_start = b1;
        b1 += q.unsplit;
    }
    gp.r2_base = b1;
    int b2 = b1;
    for (int i = 0; i < nprob; ++i) {
        GroupProblem& q = gp.pr[i];
        q.r2_start = b2;
        b2 += (q.tiles_m * q.tiles_n - q.unsplit) * q.splits;
    }
    static const bool dbg = getenv("KALLE_GEMM_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[kalle wgrad group] %d problems, %d tiles: %d whole + %d x %d slices = %d workgroups\n", nprob, ntot,
                     b1, ntot - b1, plan.splits, b2);
    if (gp.overwrite && ntot - b1 > 0 && plan.splits >= 1) {
        // (N % 4 == 0 is implied by N % 8 == 0: the clears are 16-byte stores)
        KALLE_LAUNCH(gemm3_group_zero_kernel, dim3(4 * (ntot - b1)), dim3(256), 0, static_cast<hipStream_t>(stream), gp);
        if (kalle_check_launch() != KALLE_OK) return KALLE_ERR_LAUNCH;
    }
    constexpr int lds = GEMM3_LDS;
    static std::atomic<uint64_t> lds_ok{0};
    kalle_allow_lds(reinterpret_cast<const void*>(gemm3_wgrad_group_kernel), lds, lds_ok);
    KALLE_LAUNCH(gemm3_wgrad_group_kernel, dim3(b2), dim3(512), lds, static_cast<hipStream_t>(stream), gp);
    return kalle_check_launch();
}
