/* kalle_hip.h - C-ABI of libkalle_hip.so: the MI355X (gfx950) kernels behind the kalle-audio DiT /
 * audio-VAE hot path.
 *
 * The reference (18281818221/kalle-audio) has no FFI/operator interface for this path: its seam is
 * Python nn.Module classes that call torch ops.  Each entry point below therefore
 * names the reference call site (file:line under /root/reference) whose torch op it replaces; the
 * Python drop-in modules in kalle_audio_amd/stable_audio_tools bind these with ctypes
 * (INTEGRATION.md shows the binding a maintainer would add on the reference side).
 *
 * Conventions: plain device pointers + sizes, `stream` is a hipStream_t passed as void*, every call
 * is asynchronous on that stream, allocates nothing, and returns 0 on success or a negative KALLE_ERR_* code (never
 * throws).  Process state the library does keep: per-thread caches of GEMM split plans (pure functions of the shape), the
 * calling thread's last plan / last HIP error name (kalle_gemm_last_plan, kalle_last_error), a per-kernel per-device flag for
 * the dynamic-LDS attribute, and experiment switches read once from KALLE_* environment variables.  bf16 tensors are raw uint16 storage.
 */
#ifndef KALLE_HIP_H
#define KALLE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KALLE_BF16 0
#define KALLE_F32 1

/* library / device info: returns the ABI version; fills the gfx arch name the code objects target */
int kalle_abi_version(void);
const char* kalle_target_arch(void);
/* name of the HIP error behind the calling thread's most recent KALLE_ERR_LAUNCH ("" if none) */
const char* kalle_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM (bf16 MFMA, fp32 accumulate) with fused epilogue.
 *   C[M,N] = alpha * op(A) @ op(B)  (+bias[n]) (* sigmoid(1-gate[m/rows_per_batch, n])) (+residual[m,n]) (+C)
 *   a_kmajor=0: A stored [M][K] (lda)   | a_kmajor=1: A stored [K][M] (lda)
 *   b_kmajor=0: B stored [N][K] (ldb)   | b_kmajor=1: B stored [K][N] (ldb)
 * forward  y = x @ W^T            : a_kmajor=0, b_kmajor=0  (nn.Linear: transformer.py:216,252,411,414,419,541,774,807)
 * dgrad    dx = dy @ W            : a_kmajor=0, b_kmajor=1
 * wgrad    dW = dy^T @ x          : a_kmajor=1, b_kmajor=1
 * c_dtype: KALLE_BF16 or KALLE_F32.  N, K (and M when a_kmajor) must be multiples of 8; pointers 16-B aligned.
 */
typedef struct kalle_gemm_epilogue {
    const float* bias;      /* [N] fp32 or NULL                                   (transformer.py:207,252 Linear bias) */
    const float* gate;      /* [rows/rows_per_batch][ldg] fp32 or NULL: adaLN gate (transformer.py:667,681)            */
    int64_t ldg;
    int32_t rows_per_batch;
    const float* residual;  /* [M][ldr] fp32 or NULL: residual stream add          (transformer.py:668,682,685-693)    */
    int64_t ldr;
    int32_t accumulate;     /* C += result (fp32 C only): gradient accumulation                                        */
    float alpha;            /* 0 is read as 1                                                                          */
    /* optional output-row remap (0 = off): logical row m is stored at row
     *   (m / c_rows_per_batch) * c_batch_rows + c_row_offset + m % c_rows_per_batch   of C (and of residual):
     * lets project_in write straight behind the prepended tokens of the residual stream (transformer.py:774-781) */
    int32_t c_rows_per_batch;
    int32_t c_batch_rows;
    int32_t c_row_offset;
    const uint8_t* row_mask; /* [M] or NULL: rows with 0 contribute 0 before the residual add
                              * (Attention zeroes padded query rows after to_out, transformer.py:543-545) */
    /* fused SwiGLU (transformer.py:216-219), bf16 C; in the 256x256 kernel (glu_inner % 128 == 0) and, forward only, in the
     * small-tile kernels of the few-row path (M <= 2048, glu_inner % 32 == 0); KALLE_ERR_UNSUPPORTED otherwise - the caller
     * then runs the GEMM and kalle_swiglu_* separately:
     *   glu_mode 1 (forward,  a_kmajor=0,b_kmajor=0, N = 2*glu_inner): C = h = x W^T + b  [M][2*inner]  AND
     *              glu_aux = act [M][inner] bf16 = h[:, j] * silu(h[:, inner + j])
     *   glu_mode 2 (backward, a_kmajor=0,b_kmajor=1, N = glu_inner):   acc = d(act); glu_aux = h [M][2*inner] bf16;
     *              C = dh [M][2*inner] bf16; glu_dbias (fp32 [2*inner], optional) += column sums of dh */
    int32_t glu_mode;
    int32_t glu_inner;
    void* glu_aux;
    float* glu_dbias;
    /* optional scratch for the few-rows paths (sampling at generation batch sizes, small training batches): with
     * workspace_bytes >= 8 * M * N the dispatcher may cut K into up to workspace_bytes / (4 M N) slices, each writing its
     * partial result as an fp32 [M][N] slab, and sum the slabs in a fixed order + apply the epilogue in a finishing pass
     * (bitwise reproducible).  Used for long K only: with M <= 2048 rows and N % 64 == 0 (N % 128 for a k-major B) the
     * dispatcher first cuts the OUTPUT into 64 x 64 ... 128 x 128 tiles over the whole K (two wave groups per tile, epilogue in
     * the same launch, no scratch), and takes K slices only beyond ~2048 deep.  The library never allocates: no workspace, no
     * sliced path.  Contents are undefined afterwards; one workspace per stream. */
    void* workspace;
    int64_t workspace_bytes;
} kalle_gemm_epilogue;

int kalle_gemm_bf16(const void* A, int64_t lda, int a_kmajor, const void* B, int64_t ldb, int b_kmajor,
                    void* C, int64_t ldc, int c_dtype, int M, int N, int K,
                    const kalle_gemm_epilogue* ep, void* stream);

/* which kernel the calling thread's most recent kalle_gemm_bf16 used: low byte 1 = gemm_bf16_kernel (128x128,
 * register-staged, any shape), 2 = gemm2_kernel (256x128, LDS-DMA 3-stage ring, K % 8 == 0), 3 = gemm3_kernel (256x256, 2 stages),
 * 4 = few-rows K slices (slabs + finishing pass), 5 = small tiles with two wave groups (gemm2_ks2_kernel); for 1-4 bits 8.. =
 * split-K factor, for 5 bits 8-11 / 12-15 = tile rows / columns in units of 64, bits 16.. = K slices */
int kalle_gemm_last_plan(void);

/* diagnostics (tools/gemm_stamps.py), never set by the product path: with a non-NULL device buffer of
 * [workgroups][2][8] uint64 the 256x256 kernel's wave 0 / wave 7 leave s_memrealtime stamps (100 MHz) at: 0 entry, 1 first
 * K-tile landed, 2 main loop done, 3 epilogue barrier passed, 4 stores issued, 5 stores acknowledged (an extra wait that only
 * exists while the buffer is set).  NULL switches it off again.  Process-wide. */
int kalle_gemm_debug_stamps(void* buf);
/* the same for the attention kernels (tools/attn_stamps.py): [B * H workgroups][8] stamps of thread 0 - forward: 0 entry, 1 tiles
 * staged, 2 computed, 3 stores issued, 4 stores acknowledged; fused self-attention backward: 0 entry, 1 tiles staged, 2 row
 * statistics ready, 3 dK / dV computed, 4 dK / dV stored, 5 dQ computed, 6 stores issued, 7 acknowledged */
int kalle_attn_debug_stamps(void* buf);
/* diagnostics (tools/cu_hold_probe.py), never called by the product path: `nwg` workgroups of 256 threads with `lds_bytes` of LDS
 * each (<= 64 KiB) that do nothing but stay resident for `microseconds` - a stand-in for a communication kernel that holds some
 * CUs while the GEMMs of the backward pass launch (a 160-KiB-LDS GEMM workgroup cannot share a CU with it) */
int kalle_debug_hold_cus(int nwg, int lds_bytes, int microseconds, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (bias-less gamma, eps 1e-5) with optional adaLN modulation - one wavefront per row.
 *   y = ((x-mean)*rstd*gamma + beta) * (1 + scale[b]) + shift[b]        (transformer.py:173-192, 660-665, 677-679)
 * x: [rows][D] fp32 (residual stream) or bf16; y: bf16; mean/rstd: [rows] fp32 saved for backward.
 * beta, scale, shift may be NULL.  D % 8 == 0, D <= 4096.
 */
int kalle_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta,
                        const float* scale, const float* shift, int64_t ld_mod, int rows_per_batch,
                        void* y, float* mean, float* rstd, int rows, int D, float eps, void* stream);

/* backward of the above w.r.t. x and gamma/beta.
 *   dx_out = dres + dLN/dx(dy)   (dres: incoming residual-stream gradient fp32 or NULL; may alias dx_out)
 *   dx_bf16 (optional): the same values rounded to bf16 - the operand of the next dgrad/wgrad GEMMs, saving a cast pass
 *   dgamma_part/dbeta_part: [nparts][D] fp32 partial sums (nparts = value returned by
 *   kalle_layernorm_bwd_parts(rows)); reduce with kalle_colsum_f32.
 */
int kalle_layernorm_bwd_parts(int rows);
int kalle_layernorm_bwd(const void* dy, const void* x, int x_dtype, const float* gamma,
                        const float* scale, int64_t ld_mod, int rows_per_batch,
                        const float* mean, const float* rstd, const float* dres, float* dx_out, void* dx_bf16,
                        float* dgamma_part, float* dbeta_part, int rows, int D, void* stream);

/* same, with the parameter gradients added ATOMICALLY into [D] accumulators (dgamma_acc, dbeta_acc or NULL) that the
 * caller has initialised - the trainer's flat gradient: no partial rows, no reduction launch */
int kalle_layernorm_bwd_acc(const void* dy, const void* x, int x_dtype, const float* gamma,
                            const float* scale, int64_t ld_mod, int rows_per_batch,
                            const float* mean, const float* rstd, const float* dres, float* dx_out, void* dx_bf16,
                            float* dgamma_acc, float* dbeta_acc, int rows, int D, void* stream);

/* the _acc form with the COLUMN SUMS of the bf16-rounded dx added atomically into dx_colsum_acc [D] (fp32) in place of dbeta:
 * dx of a block's pre_norm is the output gradient of the block below, whose FF-out bias gradient (transformer.py:252,
 * nn.Linear(inner, dim) with bias) is exactly that sum - fused here it saves a pass over the [rows][D] bf16 gradient */
int kalle_layernorm_bwd_colsum(const void* dy, const void* x, int x_dtype, const float* gamma,
                               const float* scale, int64_t ld_mod, int rows_per_batch,
                               const float* mean, const float* rstd, const float* dres, float* dx_out, void* dx_bf16,
                               float* dgamma_acc, float* dx_colsum_acc, int rows, int D, void* stream);

/* adaLN modulation gradients: dscale[b,d] = sum_t dy*ln, dshift[b,d] = sum_t dy   (transformer.py:665,679) */
int kalle_adaln_mod_bwd(const void* dy, const void* x, int x_dtype, const float* gamma, const float* beta,
                        const float* mean, const float* rstd, float* dscale, float* dshift, int64_t ld_mod,
                        int nbatch, int rows_per_batch, int D, void* stream);

/* RMSNorm  y = x * scale * rsqrt(mean(x^2)+eps)   (blocks.py:268-272, 285-299; AdaRMSNorm 211-221 via scale=[b]) */
int kalle_rmsnorm_fwd(const void* x, int x_dtype, const float* scale, int64_t ld_scale, int rows_per_batch,
                      void* y, int y_dtype, float* rrms, int rows, int D, float eps, void* stream);
/* dx = RMSNorm backward (+ dres, the residual-stream gradient, when given); dx_bf16: optional bf16 copy of dx for the
 * next GEMM (the Llama decoder layers under model_sigmaVAE.py:78-81: input_layernorm / post_attention_layernorm) */
int kalle_rmsnorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* scale, int64_t ld_scale,
                      int rows_per_batch, const float* rrms, float* dx, float* dscale_part, const float* dres,
                      void* dx_bf16, int rows, int D, void* stream);

/* same, with dscale added ATOMICALLY into the caller-initialised [D] accumulator dscale_acc (the trainer's flat gradient) */
int kalle_rmsnorm_bwd_acc(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* scale, int64_t ld_scale,
                          int rows_per_batch, const float* rrms, float* dx, float* dscale_acc, const float* dres,
                          void* dx_bf16, int rows, int D, void* stream);

/* Attention(qk_norm=...) (transformer.py:303-307, 422-428): every 64-wide head of q / k is normalised before the rotary
 * embedding.  mode 1 "l2": F.normalize = x / max(||x||_2, 1e-12);  mode 2 "ln": LayerNorm(64, eps 1e-6), gamma / beta fp32 [64].
 * x, y: bf16 row-major with leading dimensions ldx / ldy and element offsets (the q or k slice of a projection output, as in
 * kalle_attention_fwd; all multiples of 8); `heads` consecutive heads per row.  stat: fp32 [rows][heads][2], written by the
 * forward (mean | clamp flag, reciprocal std | reciprocal norm) and read by the backward.
 * Backward: g = gradient w.r.t. the normalised values (bf16, as kalle_attention_bwd leaves it), dx may alias g; mode 2 ADDS the
 * column sums into dgamma / dbeta (fp32 [64], atomics; either may be NULL). */
int kalle_head_norm_fwd(const void* x, int64_t ldx, int64_t x_off, void* y, int64_t ldy, int64_t y_off, float* stat,
                        const float* gamma, const float* beta, int mode, int64_t rows, int heads, void* stream);
int kalle_head_norm_bwd(const void* x, int64_t ldx, int64_t x_off, const float* stat, const void* g, int64_t ldg,
                        int64_t g_off, void* dx, int64_t lddx, int64_t dx_off, const float* gamma, float* dgamma,
                        float* dbeta, int mode, int64_t rows, int heads, void* stream);

/* column sums: out[c] (+)= sum_r in[r][c]; in fp32 or bf16 [rows][ld]. Used for bias grads and partial reduces. */
int kalle_colsum(const void* in, int in_dtype, int64_t ld, float* out, int rows, int cols, int accumulate,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise / reductions on the DiT path
 */
/* SwiGLU: out[m, j] = h[m, j] * silu(h[m, inner + j])                     (transformer.py:218-219) */
int kalle_swiglu_fwd(const void* h, void* out, int64_t rows, int inner, void* stream);
/* dh[m, j] = dout*silu(g), dh[m, inner+j] = dout*x*silu'(g);  dbias (optional, fp32 [2*inner]) += column sums of dh
 * (the GLU projection's bias gradient, fused so dh is not re-read; atomically accumulated - zero it for a fresh sum) */
int kalle_swiglu_bwd(const void* dout, const void* h, void* dh, float* dbias, int64_t rows, int inner, void* stream);
/* SiLU on fp32/bf16 vectors (to_cond_embed / to_global_embed / to_scale_shift_gate: dit.py:39-72, transformer.py:641-644) */
int kalle_silu_fwd(const void* x, void* y, int dtype, int64_t n, void* stream);
int kalle_silu_bwd(const void* dy, const void* x, void* dx, int dtype, int64_t n, void* stream);

/* forward noising + target (training/diffusion.py:365-379, inference/sampling.py:8-11)
 *   objective 0 ("v"): a=cos(pi t/2), s=sin(pi t/2); x_t = a x + s n ; target = a n - s x
 *   objective 1 ("rectified_flow"): a=1-t, s=t ;     x_t = a x + s n ; target = n - x
 * x, noise: fp32 [B][per_sample]; t fp32 [B]; x_t, target fp32.
 */
int kalle_diffuse_fwd(const float* x, const float* noise, const float* t, float* x_t, float* target,
                      int nbatch, int64_t per_sample, int objective, void* stream);

/* MSE loss (training/losses/losses.py:53-69): loss = mean((out-target)^2) over the masked elements,
 * dout = dloss * 2 (out-target)/count.  mask: uint8 [B][T] over the last dim (broadcast over C) or NULL.
 * out/target fp32 [B][C][T].  loss_sum[0] += sum of squares, loss_sum[1] += count (both must be zeroed by the caller);
 * call kalle_mse_finish to produce loss and scale dout.
 */
int kalle_mse_fwd(const float* out, const float* target, const uint8_t* mask, float* loss_acc, float* diff,
                  int nbatch, int C, int T, void* stream);
int kalle_mse_finish(float* loss_acc, float* loss, float* diff, int64_t n, float weight, void* stream);

/* batched 2-D transpose with dtype conversion (the "b c t -> b t c" rearranges of dit.py:199,219):
 *   out[b][c][r] = in[b][r][c] for r < R, c < Cn; element (b,r,c) of `in` at b*in_batch_stride + r*in_ld + c,
 *   element (b,c,r) of `out` at b*out_batch_stride + c*out_ld + r.  dtypes fp32 or bf16 independently. */
int kalle_transpose_2d(const void* in, int in_dtype, int64_t in_batch_stride, int64_t in_ld, void* out,
                       int out_dtype, int64_t out_batch_stride, int64_t out_ld, int nbatch, int R, int Cn,
                       void* stream);

/* strided row copy / cast / accumulate: out[b][r][0:cols] (+)= in[b][r][0:cols]  (cols, strides % 4 == 0).
 * Used to splice the prepended conditioning token into the residual stream (transformer.py:776-787) and to
 * gather token rows for GEMM operands. */
int kalle_copy_rows(const void* in, int in_dtype, int64_t in_batch_stride, int64_t in_ld, void* out,
                    int out_dtype, int64_t out_batch_stride, int64_t out_ld, int nbatch, int rows, int cols,
                    int accumulate, void* stream);

/* All weight gradients of a transformer block in ONE launch: dw_i [N_i][K_i] += dy_i^T x_i over `tokens` rows
 * (the autograd backward of the block's nn.Linear layers - transformer.py:216,252,411,414,419,541 - which the reference
 * leaves to torch, one GEMM each).  dy bf16 [tokens][N] (ld lddy), x bf16 [tokens][K] (ld ldx), dw fp32 [N][K] (ld lddw),
 * ACCUMULATED into (the trainer's gradient sink; zero it for a plain gradient).  N, K, lddy, ldx % 8 == 0, 16-byte aligned
 * pointers; tokens % 8 != 0 returns KALLE_ERR_UNSUPPORTED (launch the gradients one by one with kalle_gemm_bf16 then).
 * Most 256 x 256 output tiles run over all tokens (read-add-store), a tail is cut into token slices (atomic adds) so that
 * the last round of workgroups is full. */
#define KALLE_MAX_GROUP 8
typedef struct kalle_wgrad_problem {
    const void* dy;
    int64_t lddy;
    const void* x;
    int64_t ldx;
    float* dw;
    int64_t lddw;
    int32_t N, K, tokens;
} kalle_wgrad_problem;
/* overwrite != 0: dw_i = dy_i^T x_i instead (no clear needed: tiles that run whole store their result, the regions of the
 * tiles that are cut into token slices are cleared by a small launch first) */
int kalle_gemm_wgrad_group(const kalle_wgrad_problem* problems, int nprob, int overwrite, void* stream);

/* Chunked VAE encode / decode (AudioAutoencoder.encode_audio / decode_audio, autoencoders.py:429-560) as a batched pipeline:
 * every chunk rides on the batch axis of ONE encoder / decoder pass; this call is the gather in front of it and the paste
 * of the trimmed chunk centres behind it.  For segment s < nseg (<= KALLE_MAX_SEGMENTS), batch item b, row (channel) c:
 *   dst[s*dst_seg_stride + b*dst_batch_stride + c*dst_ld + dst_off[s] + j] =
 *   src[s*src_seg_stride + b*src_batch_stride + c*src_ld + src_off[s] + j]          for j < len[s]
 * offsets / strides in elements; src_off / dst_off / len are HOST arrays (copied into the launch); destinations of
 * different segments must not overlap; src and dst have the same dtype (fp32 or bf16). */
#define KALLE_MAX_SEGMENTS 64
int kalle_segment_copy(const void* src, void* dst, int dtype, int nseg, const int64_t* src_off, const int64_t* dst_off,
                       const int* len, int nbatch, int rows, int64_t src_seg_stride, int64_t src_batch_stride,
                       int64_t src_ld, int64_t dst_seg_stride, int64_t dst_batch_stride, int64_t dst_ld, void* stream);

/* dtype conversion fp32 <-> bf16 (n elements, n % 8 == 0 not required) */
int kalle_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, void* stream);

/* residual-stream gradient -> bf16 GEMM operand, with the adaLN gate backward fused in:
 *   gb[b,t,n]   = bf16( g[b,t,n] * (gate ? sigmoid(1-gate[b,n]) : 1) * (row_mask ? row_mask[b*T+t] : 1) )
 *   dgate[b,n]  = -(1 - sigmoid(1-gate[b,n])) * sum_t g[b,t,n]*row_mask * (x_out[b,t,n] - x_in[b,t,n])     (gate != NULL)
 * (x_out = x_in + branch*sigmoid(1-gate), transformer.py:667-668,681-682; x_out/x_in/dgate unused when gate == NULL).
 * D % 4 == 0, ldg % 4 == 0, 16-byte aligned rows; dgate is cleared and then summed atomically over chunks of rows. */
int kalle_grad_cast(const float* g, const float* x_out, const float* x_in, const float* gate, int64_t ldg,
                    const uint8_t* row_mask, void* gb, float* dgate, int nbatch, int rows_per_batch, int D,
                    void* stream);

/* timestep Fourier features (blocks.py:84-93): out[b, j] = cos(2 pi t[b] w[j]), out[b, F/2+j] = sin(...) ; out bf16 or fp32 */
int kalle_fourier_features(const float* t, const float* w, void* out, int out_dtype, int nbatch, int half,
                           void* stream);

/* dw[j] = sum_b 2 pi t[b] * (dout[b, half+j] cos(f) - dout[b, j] sin(f)),  f = 2 pi t[b] w[j]   (dout fp32 [B][2*half]) */
int kalle_fourier_features_bwd(const float* dout, const float* t, const float* w, float* dw, int nbatch, int half,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention (optional causal mask, optional key mask), LDS-resident K/V tiles, bf16 MFMA, fp32 softmax.
 *   out[b, i, h*64+d] = softmax_j(q_i . k_j / 8 + maskbias_j) v_j              (transformer.py:382-387, 494, 514-530)
 * q/k/v are read in place from the projection outputs (no head transposes):
 *   q: [B][Nq][ldq]  head h at column q_off + h*64 ; k: [B][Nk][ldk] at k_off + (h / (H/Hkv))*64 ; v likewise
 *   (self-attention: q,k,v all point into the fused to_qkv output, ld = 3*D, offsets 0, D, 2D;
 *    cross-attention: q from to_q (ld=D), k/v from to_kv output (ld=2*Dc, offsets 0, Dc); GQA repeat_interleave 337-340)
 * rope_cos/rope_sin: [Npos][rot/2] fp32 tables or NULL - rotary on the first `rot` dims of q and k, applied on the fly:
 *   rot = 32 the DiT's partial rotary (transformer.py:146-170, 430-444), rot = 64 the Llama decoder's (HF
 *   `apply_rotary_pos_emb`, the third-party model under model_sigmaVAE.py:17-29).
 * causal != 0: query i attends keys j <= i + (Nk - Nq) only (the Llama decoder called at model_sigmaVAE.py:78-81); the
 *   queries are then the LAST Nq positions (rotary position of query row i = i + Nk - Nq), which is what decoding against
 *   a KV cache needs (model_sigmaVAE.py:122-146 re-runs the prefix instead).
 * key_mask: uint8 [B][Nk] (1 = attend) or NULL.  lse: [B][H][Nq] fp32 saved for backward.  head dim fixed at 64.
 */
int kalle_attention_fwd(const void* q, int64_t ldq, int q_off, const void* k, int64_t ldk, int k_off,
                        const void* v, int64_t ldv, int v_off, void* out, int64_t ldo, float* lse,
                        const float* rope_cos, const float* rope_sin, int rot, const uint8_t* key_mask, int causal,
                        int B, int H, int Hkv, int Nq, int Nk, void* stream);
/* backward: dq/dk/dv written with the same strides/offsets into dq_buf/dk_buf/dv_buf (bf16). For GQA dk/dv are
 * summed over the query heads sharing a kv head; RoPE is un-rotated on dq/dk. delta scratch: [B][H][Nq] fp32. */
int kalle_attention_bwd(const void* q, int64_t ldq, int q_off, const void* k, int64_t ldk, int k_off,
                        const void* v, int64_t ldv, int v_off, const void* out, const void* dout, int64_t ldo,
                        const float* lse, float* delta, void* dq, void* dk, void* dv,
                        const float* rope_cos, const float* rope_sin, int rot, const uint8_t* key_mask, int causal,
                        int B, int H, int Hkv, int Nq, int Nk, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer: fused Adam / AdamW over a flat fp32 master buffer, also emitting the bf16 compute copy.
 *   (training/utils.py:88-93 torch.optim / FusedAdam; train_offline.py:94-100 AdamW)
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce SUM, loss scaling, ...).
 */
int kalle_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_bf16,
                    int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled,
                    int step, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * 1-D convolution stack of the audio VAEs (no MFMA; LDS line buffers, coalesced HBM).
 * Layout (B, C, L) fp32 or bf16 activations, fp32 weights (weight-norm already folded: w = g * v/||v||).
 */
/* fold weight norm and repack to the kernels' weight layout [Cin][K][CoutP] fp32, CoutP = Cout rounded up to a multiple
 * of 8 with the pad columns zeroed (w_packed must hold Cin*K*CoutP floats), so a wave's 8 output-channel weights of a
 * tap are one aligned scalar load (once per forward; weights are frozen in every reference script, factory.py:77-80):
 *   transposed=0 (Conv1d):          v [Cout=d0][Cin=d1][K], g [Cout] -> w[ci][k][co] = g[co] v[co][ci][k] / ||v[co,:,:]||
 *   transposed=1 (ConvTranspose1d): v [Cin=d0][Cout=d1][K], g [Cin]  -> w[ci][k][co] = g[ci] v[ci][co][k] / ||v[ci,:,:]||
 *   g == NULL: repack only (plain nn.Conv1d).   (dac.nn.layers.WNConv1d -> torch weight_norm; autoencoders.py:9)
 *   transposed | 2: additionally store tap k at K-1-k - with transposed=1 on a Conv1d's (v, g) this is the weight of the
 *   convolution that computes the DATA gradient of a stride-1 conv (kalle_conv1d_fwd over dy, padding (K-1)*dil - pad). */
int kalle_weight_norm_fold(const float* v, const float* g, float* w_packed, int d0, int d1, int ksize,
                           int transposed, void* stream);
/* activation descriptor of the conv kernels.  code: 0 none, 1 Snake / SnakeBeta x + sin^2(a x)/(b + 1e-9) with per-channel
 * alpha / beta (exp() applied first when logscale; plain Snake passes alpha as beta too; blocks.py:301-339,
 * backup/flows.py:51-62,113-126), 2 ELU, 3 LeakyReLU(negative_slope = param) (backup/flows.py:177-181,217,230),
 * 4 (input side only) WaveNet gate tanh(x[:, :Cin]) * sigmoid(x[:, Cin:]) - x then has 2*Cin channels
 * (backup/flows.py:614-620). */
typedef struct kalle_act {
    int32_t code;
    int32_t logscale;
    const float* alpha;
    const float* beta;
    float param;
} kalle_act;
/* what the conv kernels fuse into their store:
 *   v = (conv + bias + residual) * out_scale;  if accumulate: v += y (previous contents: sum over the parallel AMP blocks,
 *   backup/flows.py:517-523);  v = post_act(v) (the NEXT layer's input activation, per output channel, applied once here
 *   instead of once per consumer tile);  if tanh: v = tanh(v) (autoencoders.py:185).  residual has y's shape and x's dtype. */
typedef struct kalle_conv_epilogue {
    const void* residual;
    float out_scale;
    int32_t accumulate;
    int32_t tanh;
    kalle_act post_act;
    void* y_raw;   /* optional second output: the value BEFORE post_act / tanh (y's shape and dtype) - a residual unit's output
                      is needed raw by the next unit's skip path and activated by its first conv */
} kalle_conv_epilogue;
/* y = epilogue(conv1d(in_act(x), w, b, stride, padding, dilation)); in_act / epi may be NULL (= none / plain store).
 *   `padding` is the LEFT zero pad; the right pad is implied by Lout (symmetric padding: autoencoders.py:45,76;
 *   causal left-only padding: backup/flows.py:574-575,602-603).  ksize <= 16.
 *   (autoencoders.py:39-62 ResidualUnit, 64-81 EncoderBlock, 116-191 encoder/decoder stems) */
int kalle_conv1d_fwd(const void* x, int x_dtype, const float* w_packed, const float* bias, void* y, int y_dtype, int B,
                     int Cin, int Lin, int Cout, int Lout, int ksize, int stride, int padding, int dilation,
                     const kalle_act* in_act, const kalle_conv_epilogue* epi, void* stream);
/* Few positions, many channels (the C >= 512 top of the VAE; all of a single-clip decode): stride-1 conv with the roles of
 * lanes and registers swapped (lane = 4 output channels, 16 positions in registers, weights as coalesced vector loads, the
 * step's inputs as one scalar load).  Two calls: kalle_conv_pad_act writes act(x) into a zero-padded fp32 copy
 * x_padded [B][C][Lp] (`padding` leading zeros, Lp = kalle_conv_pad_len(...)), kalle_conv1d_cfirst_fwd convolves it (same
 * epilogue struct as kalle_conv1d_fwd; fp32 only).  Strided convs (dilation 1): pass phases = stride and padding =
 * ceil(padding / stride) * stride to kalle_conv_pad_act - the copy is then de-interleaved into `stride` phase rows so that a
 * tap's inputs for consecutive outputs are consecutive.  The caller owns x_padded.  Any B x C (rows ride on grid x). */
int kalle_conv_pad_len(int Lout, int ksize, int stride, int padding, int dilation);
int kalle_conv_pad_act(const float* x, float* x_padded, int B, int C, int Lin, int Lp, int padding, const kalle_act* act,
                       int phases, void* stream);
/* workspace (optional, fp32, kalle_conv_cfirst_ws_floats(...) elements; that count is 0 when it would not be used): lets the
 * launch split the input channels over workgroups too - partial sums land there and a second small kernel applies the
 * epilogue.  For the bottom of a single-clip encode (1024 -> 2048 stride 8 and 2048 -> 128 at 215 positions, the encoder of
 * autoencoders.py:116-147 as twj_dataset.py:239 calls it clip by clip). */
int kalle_conv_cfirst_ws_floats(int B, int Cin, int Cout, int Lout, int ksize);
int kalle_conv1d_cfirst_fwd(const float* x_padded, const float* w_packed, const float* bias, float* y, int B, int Cin,
                            int Lp, int Cout, int Lout, int ksize, int stride, int padding, int dilation,
                            const kalle_conv_epilogue* epi, float* workspace, void* stream);
/* the same for a transposed conv (one pass per output phase): x_padded [B][C][Lp] = kalle_conv_pad_act(x, padding =
 * ceil(ksize / stride) - 1), Lp >= kalle_convT_pad_len(Lout, ksize, stride, padding) */
int kalle_convT_pad_len(int Lout, int ksize, int stride, int padding);
int kalle_conv_transpose1d_cfirst_fwd(const float* x_padded, const float* w_packed, const float* bias, float* y, int B,
                                      int Cin, int Lp, int Cout, int Lout, int ksize, int stride, int padding,
                                      const kalle_conv_epilogue* epi, void* stream);
/* y = epilogue(conv_transpose1d(in_act(x), w, b, stride, padding))   (autoencoders.py:98-100 DecoderBlock);
 * Lout may be shorter than the full length: the causal variant trims the last `stride` outputs (backup/flows.py:383-384);
 * and up to `padding` longer (outputs the symmetric right trim would drop: the data gradient of a strided conv) */
int kalle_conv_transpose1d_fwd(const void* x, int x_dtype, const float* w_packed, const float* bias, void* y,
                               int y_dtype, int B, int Cin, int Lin, int Cout, int Lout, int ksize, int stride,
                               int padding, const kalle_act* in_act, const kalle_conv_epilogue* epi, void* stream);
/* ------------------------------------------------------------------------------------------------
 * Llasa task model head / tail (model_sigmaVAE.py:53-104); the Llama decoder layers in between run on kalle_gemm_bf16,
 * kalle_rmsnorm_*, kalle_attention_* (causal, rot = 64, GQA).
 */
/* out = a x + b y, fp32   (fixed-sigma sampling x = mean + std * randn, model_sigmaVAE.py:150-166) */
int kalle_axpby(const float* x, const float* y, float* out, float a, float b, int64_t n, void* stream);
/* one-row GEMM for decoding against a KV cache: y[n] = sum_k W[n][k] x[k] (+ residual[n]); x bf16 [K], W bf16 [N][ldw],
 * y bf16 or fp32 [N], residual fp32 [N] or NULL; K % 8 == 0, K <= 32768 */
int kalle_gemv_bf16(const void* x, const void* W, int64_t ldw, void* y, int y_dtype, const float* residual, int N, int K,
                    void* stream);
/* One KV-cached decode step of the whole Llama stack (batch 1, one new position t0), sequenced on the C side so that a
 * generated frame costs one host call instead of ~150 Python-level launches (model_sigmaVAE.py:122-146 runs
 * `self.base_model.model(inputs_embeds=...)` over the growing prefix once per frame; with a cache only the new position
 * is computed).  Per layer: [RMSNorm + fused q|k|v GEMV, k|v written straight into cache row t0] -> causal GQA attention
 * over rows 0..t0 -> [o_proj GEMV + residual] -> [RMSNorm + up|gate GEMV] -> [SwiGLU + down GEMV + residual].
 *   layers: HOST array of n_layers descriptors; weights bf16 row-major ([out][in]; wqkv = [q;k;v], wug = [up;gate]),
 *           norm weights fp32 [D], kv_cache bf16 [cache_rows][2*Hkv*64] holding un-rotated k | v of positions < t0
 *   x: fp32 [D] input embedding of position t0; out: fp32 [D] residual stream after the last layer (final norm not applied)
 *   rope_cos / rope_sin: fp32 [>= t0+1][32]; workspace: kalle_llama_decode_ws_bytes(H, Hkv, inner) bytes, 64-byte aligned */
typedef struct kalle_llama_layer {
    const float* input_norm;
    const void* wqkv;
    const void* wo;
    const float* post_norm;
    const void* wug;
    const void* wdown;
    void* kv_cache;
} kalle_llama_layer;
int kalle_llama_decode_ws_bytes(int H, int Hkv, int inner);
int kalle_llama_decode_step(const kalle_llama_layer* layers, int n_layers, const float* x, float* out, int H, int Hkv,
                            int inner, float eps, int t0, int cache_rows, const float* rope_cos, const float* rope_sin,
                            void* workspace, void* stream);
/* waveform -> int16 PCM as the inference scripts write it (infer_0723.py:293): out = int16(clamp(x / max|x|, -1, 1) * 32767);
 * peak: one fp32 of device scratch that receives max|x|; x fp32 or bf16 */
int kalle_peak_normalize_int16(const void* x, int dtype, float* peak, int16_t* out, int64_t n, void* stream);
/* out[r, :] = audio[r, :] * audio_mask[r] + table[ids[r], :] * ids_mask[r]   (embed_tokens + masked mix, :66-73);
 * table fp32 [vocab][D], audio fp32 or bf16 [rows][D], masks fp32 [rows], out fp32 */
int kalle_embed_mix_fwd(const int64_t* ids, const float* table, const void* audio, int audio_dtype, const float* ids_mask,
                        const float* audio_mask, float* out, int64_t rows, int D, int64_t vocab, void* stream);
/* daudio = dout * audio_mask (if daudio); dtable[ids[r], :] += dout[r, :] * ids_mask[r] (if dtable; fp32 atomics) */
int kalle_embed_mix_bwd(const float* dout, const int64_t* ids, const float* ids_mask, const float* audio_mask,
                        float* dtable, float* daudio, int64_t rows, int D, int64_t vocab, void* stream);
/* exact (erf) GELU, nn.GELU() default (model_sigmaVAE.py:46) */
int kalle_gelu_fwd(const void* x, void* y, int dtype, int64_t n, void* stream);
int kalle_gelu_bwd(const void* dy, const void* x, void* dx, int dtype, int64_t n, void* stream);
/* masked fixed-sigma Gaussian KL (model_sigmaVAE.py:85-95): kl[r] = sum_c (pred - label)^2 / (2 std^2) / dim;
 * sums4 (zeroed by the caller) += {sum kl*mask_a, sum mask_a, sum kl*mask_b, sum mask_b}; the two losses are
 * sums4[0]/sums4[1] and sums4[2]/sums4[3].  bwd: dpred for upstream gradients grad_a / grad_b (device scalars). */
int kalle_gauss_kl_fwd(const float* pred, const float* label, const float* mask_a, const float* mask_b, float* sums4,
                       float std, int64_t rows, int dim, void* stream);
int kalle_gauss_kl_bwd(const float* pred, const float* label, const float* mask_a, const float* mask_b,
                       const float* sums4, const float* grad_a, const float* grad_b, float* dpred, float std,
                       int64_t rows, int dim, void* stream);

/* masked two-Gaussian KL of the Stable-Audio-VAE task model (model.py:84-100):
 *   kl[r] = sum_c KL( N(m1, s1) || N(m2, exp(l2)) ) / dim,   m2 | l2 = the halves of pred [rows][2 dim] (model.py:89-90)
 * label_mode 0: label_mean / label_std [rows][dim] hold the label statistics (the transform of model.py:84 already applied
 * by the caller); label_mode 1: label_mean is the raw label [rows][2 dim] = mean | scale and std = softplus(scale) + 1e-4
 * (label_std unused).  std is multiplied by std_mult (1.25, model.py:87).  sums4 / grad_a / grad_b as kalle_gauss_kl_*;
 * dpred [rows][2 dim]. */
int kalle_gauss_kl2_fwd(const float* pred, const float* label_mean, const float* label_std, int label_mode, float std_mult,
                        const float* mask_a, const float* mask_b, float* sums4, int64_t rows, int dim, void* stream);
int kalle_gauss_kl2_bwd(const float* pred, const float* label_mean, const float* label_std, int label_mode, float std_mult,
                        const float* mask_a, const float* mask_b, const float* sums4, const float* grad_a,
                        const float* grad_b, float* dpred, int64_t rows, int dim, void* stream);

/* ---- backward of the VAE conv stacks (training the pretransform: `enable_grad`, models/factory.py:77-80; the reference
 * leaves it to torch autograd).  A fused forward unit is y = conv(act(x)) (+ residual) (-> tanh).
 * Data gradient: the forward entry points over dy with weights re-packed by kalle_weight_norm_fold -
 *   stride-1 Conv1d:   kalle_conv1d_fwd(dy, fold(v, g, 1 | 2), K, stride 1, padding (K-1)*dil - pad, dil)
 *   strided Conv1d:    kalle_conv_transpose1d_fwd(dy, fold(v, g, 1), K, stride, padding)
 *   ConvTranspose1d:   kalle_conv1d_fwd(dy, fold(v, g, 0), K, stride, padding)
 * Weight gradient (fp32, position reduction on the vector ALU, added atomically into dW which the caller zeroes):
 *   dW[cu][cv][k] += sum_{b, m} U[b, cu, m] * V[b, cv, m*stride - padding + k*dilation]
 *   Conv1d: U = dy [B][Cout][Lout], V = x [B][Cin][Lin]  -> dW in the module's [Cout][Cin][K] layout (act_on 0)
 *   ConvTranspose1d: U = x [B][Cin][Lin], V = dy [B][Cout][Lout] -> dW [Cin][Cout][K]            (act_on 1)
 *   `act` (code 0 / 1 SnakeBeta / 2 ELU) is the conv's INPUT activation, applied on the fly to x = V (act_on 0) or U (1).
 *   With the activation on V (or none), >= 16 V channels and 1 / 4 / 7 / 8 / 16 taps the LDS-staged kernel runs (the V span of a
 *   position tile staged and activated once, U as scalar operands); a transposed conv's caller passes act(x) as U with act 0. */
int kalle_conv_wgrad(const float* U, const float* V, float* dW, int B, int CU, int CV, int MU, int LV, int ksize, int stride,
                     int padding, int dilation, int act_on, const kalle_act* act, void* stream);
/* dx = g * act'(x) for x, g [B][C][L] fp32; SnakeBeta (blocks.py:301-339) also adds d alpha, d beta [C] atomically
 * (through the exp when logscale); act codes 0 / 1 / 2 */
int kalle_act_bwd(const float* x, const float* g, float* dx, const kalle_act* act, float* dalpha, float* dbeta, int B, int C,
                  int L, void* stream);
/* g = dy * (1 - y^2): the decoder's final tanh (autoencoders.py:185) */
int kalle_tanh_bwd(const float* dy, const float* y, float* g, int64_t n, void* stream);
/* nn.Upsample(scale_factor=scale, mode="nearest") along the last axis (DecoderBlock use_nearest_upsample, autoencoders.py:87-89):
 * backward == 0: x [rows][L] -> y [rows][L*scale], y[r][l] = x[r][l / scale];
 * backward != 0: x = dy [rows][L*scale] -> y = dx [rows][L], dx[r][m] = sum_{j<scale} dy[r][m*scale + j].  fp32, scale 1..64 */
int kalle_upsample_nearest(const float* x, float* y, int64_t rows, int L, int scale, int backward, void* stream);
/* out[c] += sum_{b, l} x[b][c][l]  (bias gradient; out zeroed or accumulated by the caller) */
int kalle_channel_sum(const float* x, float* out, int B, int C, int L, void* stream);
/* torch weight_norm (dim 0) backward: w = g v / ||v|| per slice of dim 0 (n = elements per slice):
 * dg = <dw, v> / ||v||, dv = g / ||v|| (dw - v <dw, v> / ||v||^2); accumulate != 0 adds to dv / dg */
int kalle_weight_norm_bwd(const float* dw, const float* v, const float* g, float* dv, float* dg, int d0, int n,
                          int accumulate, void* stream);

/* anti-aliased periodic activation (alias-free-torch `Activation1d`, third-party, used by the mel-VAE decoder,
 * backup/flows.py:266-279,452-456): 2x kaiser-sinc FIR upsample (12 taps, replicate pad) -> x + sin^2(x a)/(b+1e-9)
 * -> 2x FIR low-pass downsample.  x, y: (B, C, L) fp32 or bf16; filter12: the 12 fp32 taps.  alpha == beta == NULL: ELU in
 * place of the snake (Oobleck units built with antialias_activation=True and use_snake=False, autoencoders.py:24-37). */
int kalle_act1d_fwd(const void* x, void* y, int dtype, const float* filter12, const float* alpha, const float* beta,
                    int logscale, int B, int C, int L, void* stream);
/* standalone SnakeBeta: y = x + sin^2(x e^alpha) / (e^beta + 1e-9)     (blocks.py:301-339) */
int kalle_snake_beta_fwd(const void* x, void* y, int dtype, const float* alpha, const float* beta, int logscale,
                         int B, int C, int L, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Off-default options of the DiT's transformer (none is set by the configs the reference ships)
 */
/* x[b][i] += table[i], b < nbatch, i < n: the position-embedding add of ContinuousTransformer(use_sinusoidal_emb /
 * use_abs_pos_emb), transformer.py:796-797 (`x = x + self.pos_emb(x)`, one [seq][dim] table for every batch element).
 * fp32, n % 4 == 0, 16-byte aligned; in place on the residual stream */
int kalle_add_rows(float* x, const float* table, int64_t nbatch, int64_t n, void* stream);
/* depthwise convolution along the sequence of token-major activations - ConformerModule.depthwise_conv, transformer.py:564
 * (Conv1d(dim, dim, 17, groups = dim, padding = 8, bias = False) between two rearranges b n d <-> b d n, 576-578):
 *   y[b][n][c] = sum_k w[c][flip ? K-1-k : k] * x[b][n + k - pad][c]        (x = 0 outside 0 <= n + k - pad < N)
 * x: bf16 [B][N][D]; w: fp32 [D][K] (the parameter's [D][1][K]); y: fp32 or bf16 [B][N][D].  flip != 0 with
 * pad = K - 1 - padding is the data gradient (x = dy).  D even, K <= 32, B <= 65535 */
int kalle_dwconv1d_fwd(const void* x, const float* w, void* y, int y_dtype, int B, int N, int D, int K, int pad, int flip,
                       void* stream);
/* its weight gradient: dw[c][k] += sum_{b, n} dy[b][n][c] * x[b][n + k - pad][c]   (dy, x bf16; dw fp32 [D][K], atomically
 * ADDED into: zero it for a fresh sum) */
int kalle_dwconv1d_wgrad(const void* dy, const void* x, float* dw, int B, int N, int D, int K, int pad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KALLE_HIP_H */
