// Shared between the two GEMM kernels (gemm.hip: 128x128 register-staged, any shape; gemm2.hip: 256-wide tiles,
// LDS-DMA pipeline, split-K) - parameter block and the fused epilogue on 4 consecutive output columns.
#pragma once
#include "common.h"

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    const float* bias;
    const float* gate;
    int64_t ldg;
    int rows_per_batch;
    const float* residual;
    int64_t ldr;
    int accumulate;
    int tiles_n;
    float alpha;
    int c_rpb, c_brows, c_roff;
    const uint8_t* row_mask;
    // gemm2 only
    int tiles_m, splits, ktiles_per_split, atomic, group_m, tile_n;
    // mixed split-K (gemm3 weight gradients): the first mix_na tiles are cut into mix_sa K slices, the rest into mix_sa + 1;
    // 1-D grid, the longer slices first, so that the last round of workgroups is made of short ones (mix_na < 0: off)
    int mix_na, mix_sa;
    // fused SwiGLU (gemm3 only): 1 = forward (C = h [M][2*inner], glu_aux = act [M][inner]); 2 = backward (acc = dact,
    // glu_aux = h, C = dh [M][2*inner], glu_dbias += column sums of dh)
    int glu_mode, glu_inner;
    void* glu_aux;
    float* glu_dbias;
    // few-rows split-K (gemm2 only): K slice s stores its partial tile to C + s * slab_stride (plain stores, summed in a fixed
    // order by the finishing pass: deterministic, and plain stores run ~4x the rate of float atomics); 0: off
    int64_t slab_stride;
    // diagnostics (tools/gemm_stamps.py; NULL in every product call): [workgroup][2 waves][8] s_memrealtime stamps (100 MHz) of
    // the 256 x 256 kernel's phases - entry, first tile landed, main loop done, epilogue barrier passed, stores issued, stores done
    unsigned long long* stamps;
    // persistent 256 x 256 kernel: every second group of 8 workgroups starts this many 10-ns ticks late, so that the CUs' epilogue
    // bursts (HBM-bound when all 256 fall together) interleave with the other half's main loops (0: lockstep)
    int dephase_ticks;
    // persistent 256 x 256 kernel, dynamic tile hand-out: 0 = every workgroup walks tiles b, b + grid, ... (static); n >= 1 = after
    // its first tile a workgroup draws the next one from counter set n - 1 (gemm2.hip, kalle_gemm_sched)
    int sched_set;
    int dbg;        // knock-out timing of the fused SwiGLU backward epilogue (KALLE_GEMM_DBG): 1 = no h loads, 2 = no dh stores
};

// Workgroup id -> output tile.  (1) XCD-aware: blocks b, b+8, ... share an XCD (and its 4 MiB L2), so each XCD gets a
// contiguous run of tile ids (bijective for any grid size).  (2) Grouped raster: consecutive ids walk GM tile-rows x
// n tile-columns column by column, so the ~32-64 tiles an XCD runs concurrently form a GM x (32/GM) patch that shares
// GM A-panels and 32/GM B-panels through L2 instead of 1 A-panel and 32 B-panels.
__device__ __forceinline__ void gemm_tile_coords(int bid, int nwg, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int per_group = gm * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * gm;
    const int gsz = min(gm, tiles_m - first_m);
    const int in = wg - group * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

// block id -> position in the work order such that the blocks of one XCD (ids b, b + 8, ...) cover a contiguous run of it
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
    const int qd = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
}

// the grouped raster alone (workgroup ids of a grouped launch are not a multiple of 8 per problem: no XCD remap there)
__device__ __forceinline__ void gemm_tile_coords_plain(int wg, int tiles_m, int tiles_n, int gm, int& tm, int& tn) {
    const int per_group = gm * tiles_n;
    const int group = wg / per_group;
    const int first_m = group * gm;
    const int gsz = min(gm, tiles_m - first_m);
    const int in = wg - group * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

// logical output row -> stored row (project_in writes behind the prepended tokens of the residual stream)
__device__ __forceinline__ int64_t gemm_crow(const GemmParams& p, int gm) {
    return p.c_rpb > 0 ? (int64_t)(gm / p.c_rpb) * p.c_brows + p.c_roff + gm % p.c_rpb : gm;
}

// v[0..3] = accumulators of row gm, columns gn..gn+3 ; applies alpha, bias, adaLN gate, row mask, residual
__device__ __forceinline__ void gemm_epilogue4(const GemmParams& p, int gm, int gn, int64_t crow, float (&v)[4],
                                               bool with_residual = true, const f32x4* bias_pre = nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] *= p.alpha;
    if (p.bias) {   // bias_pre: the caller already holds this lane's 4 bias values
        const f32x4 b = bias_pre ? *bias_pre : *reinterpret_cast<const f32x4*>(p.bias + gn);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b[j];
    }
    if (p.gate) {  // adaLN gating, transformer.py:667,681: x * sigmoid(1 - gate)
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gate + (int64_t)(gm / p.rows_per_batch) * p.ldg + gn);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= sigmoidf_(1.f - g[j]);
    }
    if (p.row_mask && !p.row_mask[gm]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 0.f;
    }
    if (with_residual && p.residual) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p.residual + crow * p.ldr + gn);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[j];
    }
}

int kalle_gemm_v1_launch(const GemmParams& p, bool a_km, bool b_km, bool f32, hipStream_t st);
int kalle_gemm_v2_launch(GemmParams& p, bool a_km, bool b_km, bool f32, hipStream_t st);  // KALLE_ERR_UNSUPPORTED if n/a
