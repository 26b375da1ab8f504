/*
 * unsloth_amd.h -- C ABI of libunsloth_amd.so (MI355X / gfx950 hot-path kernels).
 *
 * This is the drop-in boundary of the project. The reference
 * (unslothai/unsloth) has no C FFI of its own: its hot path is Triton kernels launched from
 * Python plus ONE ctypes binding, to bitsandbytes' C library (unsloth/kernels/utils.py:266-284).
 * This header therefore declares
 *   (1) the bitsandbytes symbols that binding uses, with bitsandbytes' exact C signatures, and
 *   (2) one `uamd_*` entry point per Triton kernel launch of the unsloth/kernels python modules, following the
 *       calling convention the reference already uses for native code (utils.py:198-202,242-253):
 *       raw device pointers, C int / int64 scalars, the CURRENT stream of the tensor's device
 *       passed as an opaque pointer, no allocation, no synchronisation.
 *
 * Conventions
 *   - every `uamd_*` function returns 0 on success, a negative UAMD_ERR_* code for rejected
 *     arguments, or a positive hipError_t from the launch. Nothing is launched on error.
 *   - `dtype` arguments: UAMD_F32 / UAMD_F16 / UAMD_BF16.
 *   - strides are in ELEMENTS, sizes in elements unless stated.
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); kernels are enqueued,
 *     never synchronised. All pointers are device pointers.
 *   - in-place contracts of the reference are part of the ABI and are stated per function.
 */
#ifndef UNSLOTH_AMD_H
#define UNSLOTH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAMD_OK 0
#define UAMD_ERR_DTYPE (-1)
#define UAMD_ERR_ARG (-2)
#define UAMD_ERR_ALIGN (-3)

#define UAMD_F32 0
#define UAMD_F16 1
#define UAMD_BF16 2

/* library / ABI version: (major << 16) | minor */
int uamd_version(void);

/* ---------------------------------------------------------------------------------------------
 * RMSNorm.  Replaces unsloth/kernels/rms_layernorm.py:21-59 (_rms_layernorm_forward),
 * :123-159 (_gemma_rms_layernorm_forward), :62-120 (_rms_layernorm_backward), as launched by
 * Fast_RMS_Layernorm.forward/backward (:162-240).
 *   fwd: r[row] = rsqrt(mean(x^2) + eps) (fp32);  y = (x*r).to(W.dtype) * W   (gemma: x*r*(W+1) in fp32)
 *   bwd: dX = r/n * (n*dY*W - xhat * sum(dY*W*xhat));  dX MAY ALIAS dY (the reference writes in
 *        place for the non-gemma case, :92-95,218).
 */
int uamd_rms_layernorm_fwd(const void* X, const void* W, void* Y, float* r, int64_t n_rows,
                           int n_cols, int64_t x_row_stride, int64_t y_row_stride, float eps,
                           int gemma, int x_dtype, int w_dtype, void* stream);
int uamd_rms_layernorm_bwd(const void* dY, void* dX, const void* X, const void* W, const float* r,
                           int64_t n_rows, int n_cols, int64_t dy_row_stride,
                           int64_t dx_row_stride, int64_t x_row_stride, int gemma, int x_dtype,
                           int w_dtype, void* stream);
/* Weight gradient of the norm, dW[c] (+)= sum_rows dY[row, c] * X[row, c] * r[row] (the same for Llama's w and Gemma's
 * 1 + w). The reference's backward returns none (rms_layernorm.py:218-240: norm weights are frozen under LoRA; with
 * full_finetuning=True, vision.py:2206-2209 train_layernorms, HF's torch RMSNorm + autograd compute it). Deterministic
 * two-stage column reduction; workspace: fp32 scratch, ws_elems >= n_cols (row chunks = ws_elems / n_cols, capped).
 * Call BEFORE uamd_rms_layernorm_bwd when that one overwrites dY. X is the norm's input (H for the fused add form). */
int uamd_rms_layernorm_dw(const void* dY, const void* X, const float* r, void* dW, float* workspace, int64_t ws_elems,
                          int64_t n_rows, int n_cols, int64_t dy_row_stride, int64_t x_row_stride, int accumulate,
                          int x_dtype, int w_dtype, void* stream);

/* Residual add fused into the norm (llama.py:823-844 runs `residual + x` and the norm as two passes):
 *   fwd: h = X + Res -> H (one rounding to the activation dtype), Y = rmsnorm(h) * W, r as above. H may alias X / Res.
 *   bwd: dX = rmsnorm_backward(dY; H, W, r) + dRes (both rounded like the separate ops). dX may alias dY / dRes.
 * Llama-style norm only; rows up to 64 * 8 16-byte vectors, 16-byte aligned, else UAMD_ERR_ALIGN. */
int uamd_add_rms_layernorm_fwd(const void* X, const void* Res, const void* W, void* H, void* Y, float* r,
                               int64_t n_rows, int n_cols, int64_t x_row_stride, int64_t res_row_stride,
                               int64_t h_row_stride, int64_t y_row_stride, float eps, int x_dtype, int w_dtype,
                               void* stream);
int uamd_add_rms_layernorm_bwd(const void* dY, const void* dRes, void* dX, const void* H, const void* W,
                               const float* r, int64_t n_rows, int n_cols, int64_t dy_row_stride,
                               int64_t dres_row_stride, int64_t dx_row_stride, int64_t h_row_stride, int x_dtype,
                               int w_dtype, void* stream);

/* LayerNorm (vision towers).  Replaces unsloth/kernels/layernorm.py:25-65 (layernorm_forward) and :68-104
 * (layernorm_backward), as launched by Fast_Layernorm (:107-163) behind `fast_layernorm`:
 *   fwd: mean, r = rsqrt(mean((x - mean)^2) + eps) in fp32; y = ((x - mean) r) W + b, ONE rounding to x's dtype; r and
 *        mean (fp32 [n_rows]) are kept for the backward.
 *   bwd: dX = (dY W - mean(dY W) - xhat mean(dY W xhat)) r, written IN PLACE over dY (:104); no dW / db (frozen norms).
 * W and b share w_dtype (x's dtype or fp32). */
int uamd_layernorm_fwd(const void* X, const void* W, const void* B, void* Y, float* r, float* mu, int64_t n_rows,
                       int n_cols, int64_t x_row_stride, int64_t y_row_stride, float eps, int x_dtype, int w_dtype,
                       void* stream);
int uamd_layernorm_bwd(void* dY, const void* X, const void* W, const float* r, const float* mu, int64_t n_rows,
                       int n_cols, int64_t dy_row_stride, int64_t x_row_stride, int x_dtype, int w_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * RoPE (rotate-half), IN PLACE.  backward != 0 negates sin (rope_embedding.py:140-142).
 * uamd_rope_embedding    replaces _rope_embedding    (rope_embedding.py:104-166) as launched by
 *                        Fast_RoPE_Embedding (:169-261): Q is [n_rows, n_heads*head_dim],
 *                        position = row % seqlen.
 * uamd_rope_embedding_qk replaces _rope_embedding_QK (rope_embedding.py:23-98) as launched by
 *                        Fast_RoPE_Embedding_QK (:283-399): Q [B,Hq,T,D] and K [B,Hk,T,D] given by
 *                        element strides; rope_indices (int32 [B*T]) may be NULL (-> row % seqlen).
 * cos/sin: [>=max_pos, >=head_dim/2] tables, only the first head_dim/2 columns are read.
 */
int uamd_rope_embedding(void* Q, int64_t q_row_stride, const void* cos, int64_t cos_row_stride,
                        const void* sin, int64_t sin_row_stride, int64_t n_rows, int seqlen,
                        int n_heads, int head_dim, int backward, int q_dtype, int table_dtype,
                        void* stream);
int uamd_rope_embedding_qk(void* Q, int64_t q_batch_stride, int64_t q_head_stride,
                           int64_t q_seq_stride, void* K, int64_t k_batch_stride,
                           int64_t k_head_stride, int64_t k_seq_stride, const void* cos,
                           int64_t cos_row_stride, const void* sin, int64_t sin_row_stride,
                           const int32_t* rope_indices, int batch, int seqlen, int n_heads_q,
                           int n_heads_k, int head_dim, int backward, int q_dtype,
                           int table_dtype, void* stream);

/* Multimodal RoPE ("mrope", Qwen2-VL text tower; SURVEY 8 f4 / BASELINE config 4). The reference has no kernel for it
 * (its VLM path is the unsloth_zoo compiler); semantics = transformers' apply_multimodal_rotary_pos_emb:
 * positions3 = int32 [3, batch*seqlen] (temporal, height, width position per token); rotary pair j uses the temporal
 * position for j < section_t, the height position for j < section_t + section_h, else the width position.
 * Everything else as uamd_rope_embedding_qk (in place, strided views, backward != 0 negates sin). */
int uamd_rope_embedding_qk_mrope(void* Q, int64_t q_batch_stride, int64_t q_head_stride,
                                 int64_t q_seq_stride, void* K, int64_t k_batch_stride,
                                 int64_t k_head_stride, int64_t k_seq_stride, const void* cos,
                                 int64_t cos_row_stride, const void* sin, int64_t sin_row_stride,
                                 const int32_t* positions3, int section_t, int section_h, int batch,
                                 int seqlen, int n_heads_q, int n_heads_k, int head_dim, int backward,
                                 int q_dtype, int table_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gated MLP activations over n contiguous elements.
 * forward : h = f(e).to(dtype) * g            swiglu.py:27-47, geglu.py:31-53, :142-167
 * backward: IN PLACE  DW <- h = f*g,  e <- df = DW*f,  g <- de = DW*g*f'(e)
 *                                            swiglu.py:67-109, geglu.py:74-123, :188-244
 */
int uamd_swiglu_fg(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream);
int uamd_swiglu_DWf_DW_dfg(void* DW, void* e, void* g, int64_t n, int dtype, void* stream);
int uamd_geglu_exact_forward(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream);
int uamd_geglu_exact_backward(void* DW, void* e, void* g, int64_t n, int dtype, void* stream);
int uamd_geglu_approx_forward(const void* e, const void* g, void* h, int64_t n, int dtype, void* stream);
int uamd_geglu_approx_backward(void* DW, void* e, void* g, int64_t n, int dtype, void* stream);
/* QuickGELU y = x * sigmoid(1.702 x) (Qwen2-VL vision MLP; BASELINE config 4): forward y from x; backward dx written IN PLACE
 * over dy (dy_dx), x unchanged. fp32 arithmetic, one rounding. */
int uamd_quick_gelu_forward(const void* x, void* y, int64_t n, int dtype, void* stream);
int uamd_quick_gelu_backward(const void* x, void* dy_dx, int64_t n, int dtype, void* stream);
/* The gated activation fused with the skinny LoRA products that would re-read its output (fast_lora.py:93-96: h = f(e) * g,
 * then h @ A_down^T; :157, :172-189: h, df, de, then df @ B_up and de @ B_gate). act: 0 SwiGLU, 1 GeGLU exact, 2 GeGLU tanh;
 * element-wise results bit-identical to the plain entry points above. e / g / h (DW) are [M, K] with row stride ld;
 * W* are the LoRA factors as [R, K] with K contiguous (A_down; B_up^T, B_gate^T), R <= 64. out*: fp32 [M, ld_out], columns
 * [R, out_cols) zero-filled; out_k* (may be NULL): the same sums rounded to `dtype` -- the rank-block operand
 * uamd_gemm_group.lora_xk at its column offset -- columns [R, k_cols) zero-filled. K % 8 == 0, ld % 8 == 0. */
int uamd_glu_fwd_xa(int act, const void* e, const void* g, void* h, int M, int K, int64_t ld, const void* W, int64_t ldw,
                    int R, float* out, int64_t ld_out, int out_cols, void* out_k, int64_t ld_k, int k_cols, int dtype,
                    void* stream);
int uamd_glu_bwd_xa(int act, void* DW, void* e, void* g, int M, int K, int64_t ld,
                    const void* Wu, int64_t ldwu, int Ru, float* out_u, int64_t ld_out_u, int out_cols_u,
                    void* out_k_u, int64_t ld_k_u, int k_cols_u,
                    const void* Wg, int64_t ldwg, int Rg, float* out_g, int64_t ld_out_g, int out_cols_g,
                    void* out_k_g, int64_t ld_k_g, int k_cols_g, int dtype, void* stream);
/* The same two calls with the columns of every 16-row group SPLIT into TWO EVEN parts taken by adjacent workgroups -- where that
 * measured faster (round 5, tools/probes/tile_shape_probe.hip: workgroups that each walk along their own rows sweep the matrix
 * column slab by column slab, 4.4-4.8 TB/s on the access pattern alone): rows of whole 4 KiB pages (K * itemsize % 4096 == 0) or
 * at most 2048 rows, and at least 8 column tiles; everywhere else (and for UAMD_TUNE_GLU_XA < 3, a GeGLU activation, or a target
 * other than gfx942 / gfx950) these calls run the unsplit kernels above and `ws` is unused. The rank products of a part go to `ws`
 * as fp32 partials and the workgroup that finishes a row group last adds them in part order (deterministic).
 * ws: >= uamd_glu_xa_workspace(M, K, n_products, max rank) floats -- an UPPER bound sized for parts of at least 4 tiles, not the
 * shipped two-part rule; counters: (M + 15) / 16 ints, ZERO on entry, left zero by a launch that completes (re-zero them after any
 * launch that returned an error). One workspace per device and stream.
 * Same results: the element-wise outputs bit for bit, the rank products up to fp32 summation order. */
int64_t uamd_glu_xa_workspace(int M, int K, int n_products, int max_rank);
int uamd_glu_fwd_xa_ws(int act, const void* e, const void* g, void* h, int M, int K, int64_t ld, const void* W, int64_t ldw,
                       int R, float* out, int64_t ld_out, int out_cols, void* out_k, int64_t ld_k, int k_cols,
                       float* ws, int64_t ws_floats, int* counters, int dtype, void* stream);
int uamd_glu_bwd_xa_ws(int act, void* DW, void* e, void* g, int M, int K, int64_t ld,
                       const void* Wu, int64_t ldwu, int Ru, float* out_u, int64_t ld_out_u, int out_cols_u,
                       void* out_k_u, int64_t ld_k_u, int k_cols_u,
                       const void* Wg, int64_t ldwg, int Rg, float* out_g, int64_t ld_out_g, int out_cols_g,
                       void* out_k_g, int64_t ld_k_g, int k_cols_g, float* ws, int64_t ws_floats, int* counters,
                       int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Cross entropy.  Replaces cross_entropy_loss.py:35-111 / :114-199 (+ host logsumexp :366-370) and
 * :202-285, as launched by Fast_CrossEntropyLoss (:288-418).  No vocabulary-size limit.
 *   forward : logsumexp[row], loss[row] = logsumexp - x[label]  (0 when label == -100)
 *   backward: logits <- dloss[row] * (softmax - onehot) (x scale, x (1 - tanh^2) for softcap),
 *             IN PLACE over logits (:276, :413-418); rows with label == -100 become zeros.
 * logit_softcapping / logit_scaling == 0 disable the transform. labels are int64.
 */
int uamd_cross_entropy_forward(const void* logits, int64_t logits_row_stride, float* loss,
                               float* logsumexp, const int64_t* labels, int64_t n_rows,
                               int vocab_size, float logit_softcapping, float logit_scaling,
                               int dtype, void* stream);
int uamd_cross_entropy_backward(void* logits, int64_t logits_row_stride, const float* dloss,
                                int64_t dloss_stride, const float* logsumexp,
                                const int64_t* labels, int64_t n_rows, int vocab_size,
                                float logit_softcapping, float logit_scaling, int dtype,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * bitsandbytes-compatible symbols (exact bitsandbytes signatures; bound by the reference at
 * unsloth/kernels/utils.py:272-275 and called at :650-675).  `code` may be NULL for the *_nf4
 * functions (built-in NF4 table), as the reference passes NULL (:663-664).
 */
void cdequantize_blockwise_fp32(float* code, unsigned char* A, float* absmax, float* out,
                                int blocksize, const int n, void* stream);
void cdequantize_blockwise_bf16_nf4(float* code, unsigned char* A, float* absmax, void* out,
                                    int blocksize, const int n, void* stream);
void cdequantize_blockwise_fp16_nf4(float* code, unsigned char* A, float* absmax, void* out,
                                    int blocksize, const int n, void* stream);
void cdequantize_blockwise_fp32_nf4(float* code, unsigned char* A, float* absmax, void* out,
                                    int blocksize, const int n, void* stream);

/* Native NF4 entry points: the whole of fast_dequantize (utils.py:567-679) in one launch.
 *   absmax_f32[k] = code2[absmax_u8[k]] * absmax2[k / blocksize2] + offset        (:650-659)
 *   W[j]          = NF4[nibble_j] * absmax_f32[j / blocksize]                     (:662-675)
 * Give EITHER absmax_f32 (not nested / pre-dequantised) OR the nested triple.
 * transpose_out != 0 writes out[c * ld_out + r] (W^T, for the contraction over `out` in dX = dY @ W).
 */
int uamd_dequantize_absmax(const float* code2, const uint8_t* absmax_u8, const float* absmax2,
                           float offset, float* out, int blocksize2, int64_t n, void* stream);
int uamd_nf4_dequantize(const uint8_t* packed, const float* absmax_f32, const uint8_t* absmax_u8,
                        const float* code2, const float* absmax2, float offset, int blocksize2,
                        const float* nf4_lut, void* out, int64_t rows, int64_t cols, int blocksize,
                        int out_dtype, int transpose_out, int64_t ld_out, void* stream);
/* The row-major decode of up to FOUR weights in one launch: the members of one grouped GEMM (q | k | v, gate | up), whose
 * single decodes are latency-bound ([1024, 4096]: 1.9 TB/s, [4096, 4096]: 4.4 TB/s against 5.6 for the MLP weights). Every
 * weight: first-level absmax already in fp32 (uamd_dequantize_absmax), numel a multiple of 8192, 16-bit contiguous output,
 * its own 16-entry level table (nf4_lut NULL or an entry NULL: the NF4 levels).
 * Bit-identical to uamd_nf4_dequantize (the reference decodes weight by weight: kernels/utils.py:650-675). */
int uamd_nf4_dequantize_multi(int nseg, const uint8_t* const* packed, const float* const* absmax_f32,
                              void* const* out, const int64_t* numel, const float* const* nf4_lut, int blocksize,
                              int out_dtype, void* stream);
/* First-level NF4 quantiser (absmax + 4-bit codes); replaces bitsandbytes quantize_4bit for
 * building checkpoints without bitsandbytes. blocksize: power of two in [8,512]. */
int uamd_nf4_quantize(const void* in, uint8_t* packed, float* absmax, int64_t n, int blocksize,
                      int in_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LoRA / QLoRA linear GEMMs.  Replace matmul_lora (utils.py:1128-1170) and the dX products of
 * LoRA_MLP / LoRA_QKV / LoRA_W .backward (fast_lora.py:156,193-204,497-517,639-647).
 *
 *   C_g[M,N_g] (+)= A[M,K] @ B_g[N_g,K]^T  +  lora_scale_g * T(XA_g[M,R_g]) @ LB_g[N_g,R_g]^T
 *
 * for up to 3 groups g sharing the activation A (q/k/v, gate/up). fp32 MFMA accumulation, one
 * rounding to the activation dtype. `accumulate` != 0 adds into the existing C (dX += ...).
 * uamd_gemm_nt     : B_g dense [N_g,K] (activation dtype).
 * uamd_gemm_nt_nf4 : B_g = bitsandbytes NF4 packed bytes of the [N_g,K] weight, blocksize 64,
 *                    `absmax` = fp32 statistics (uamd_dequantize_absmax); K % 64 == 0.
 * uamd_lora_xa     : XA[M,out_cols] = X[M,K] @ A[R,K]^T in fp32 (columns >= R zero-filled).
 */
typedef struct {
    const void* B;
    void* C;
    const float* absmax;
    const float* lora_xa;
    const void* lora_b;
    int64_t ldb, ldc, ld_xa, ld_lb;
    int N;
    int R;
    float lora_scale;
    int _pad;
    /* Rank block as EXTRA K TILES (uamd_gemm_nt_256 only; takes precedence over lora_xa/lora_b there):
     *   C_g += XK[M, Rk] @ BK_g[N_g, Rk]^T   contracted in the same LDS-DMA pipeline as A @ B_g^T,
     * with XK = T(X A^T) (uamd_lora_xa2k: the reference's rounding point, utils.py:1166-1168), zero-padded to Rk
     * columns, and BK_g = T(lora_scale_g * LB_g) placed at its rank columns of a zero [N_g, Rk] buffer
     * (uamd_lora_prepare dst_pad). Rk % 64 == 0; NULL lora_xk = no rank block. */
    const void* lora_xk;
    const void* lora_bk;
    int64_t ld_xk, ld_bk;
    int Rk;
    int _pad2;
    /* bias of the base layer (activation dtype, [N_g]) added in the epilogue before the single rounding, or NULL.
     * Qwen2-style q/k/v projections carry one; the reference falls back to the un-fused PEFT path for them
     * (unsloth/models/llama.py:3695-3772). Pass it on the forward launch only (never with accumulate != 0 chains). */
    const void* bias;
} uamd_gemm_group;

int uamd_gemm_nt(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                 int n_groups, int accumulate, int dtype, void* stream);
/* uamd_gemm_nt_256: same contract as uamd_gemm_nt with 256x256x64 tiles (128x256x64 when the launch is too small to
 * fill 256 CUs with the big tile), LDS-DMA staging and two wave groups in anti-phase (csrc/gemm256.hip). Requires
 * K % 64 == 0 and every operand to span less than 4 GiB. */
int uamd_gemm_nt_256(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                     int n_groups, int accumulate, int dtype, void* stream);
/* uamd_gemm_nn_256: the same kernels with B_g given as [K, N_g] (N contiguous, ldb = row stride) and the rank block's
 * BK_g as [Rk, N_g]: C_g (+)= A @ B_g (+ XK @ BK_g). The dX products of the backward (dX = dY @ W with W stored
 * [out, in], fast_lora.py:156, :193-204) contract over the weight's ROWS: they read the forward's row-major decode
 * through transposing LDS reads instead of a transposed copy of W. N_g % 8 == 0, ldb % 8 == 0. */
int uamd_gemm_nn_256(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                     int n_groups, int accumulate, int dtype, void* stream);
/* uamd_gemm_tn_256: A given as [K, M] (M contiguous, lda = row stride) AND B_g as [K, N_g]: C_g (+)= A^T @ B_g. The
 * weight gradient of a TRAINABLE dense projection (full fine-tuning, BASELINE config 3; the reference leaves it to
 * torch.nn.Linear's autograd: loader.py:487-523 -> FastModel): dW[out, in] (+)= dY[T, out]^T @ X[T, in], both operands
 * read where the backward left them (transposing LDS reads on both sides, no transposed copy). accumulate != 0 adds
 * into C (gradient accumulation; the row-chunked lm_head gradient of the fused linear cross entropy).
 * M % 8 == 0, N_g % 8 == 0, K % 64 == 0, lda % 8 == 0, ldb % 8 == 0; no LoRA fields. */
int uamd_gemm_tn_256(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                     int n_groups, int accumulate, int dtype, void* stream);
/* process-wide kernel-variant hooks (`uamd_set_tuning(knob, value)`; defaults are the measured-fastest values). Three of them can
 * also be set from the environment -- UAMD_ATTN_VAR, UAMD_GLU_XA, UAMD_GEMM_S: the ones whose best value depends on the workload --;
 * the A/Bs of the others are settled and they remain here for the parity tests only (every listed value is exercised by one).
 *   UAMD_TUNE_GROUP_M     row panels per raster group of the 256x256 kernel (L2 reuse) */
#define UAMD_TUNE_GLU_VAR 0     /* gated-MLP activation kernels: 0 = 2048-block grid-stride, 1 = uncapped grid,
                                 * one 16-byte vector per thread, 2 = uncapped grid, two vectors per thread */
#define UAMD_TUNE_GROUP_M 1
#define UAMD_TUNE_STREAM_NT 2   /* streaming kernels: bit0 non-temporal loads, bit1 n.t. stores */
#define UAMD_TUNE_DEQUANT_T 3   /* transposing NF4 dequant: 1 = 64x256 tile kernel, 0 = 64x64, 2 = 64x256 with
                                 * the row tile as the fastest grid index (adjacent output segments written together) */
#define UAMD_TUNE_ATTN_VAR 4    /* (UAMD_ATTN_VAR) attention forward: 0 = by shape (plain causal batches with >= 2 work items per CU take
                                 * attn_fwd_ps_kernel -- one persistent workgroup per CU, ping-pong wave groups -- everything else
                                 * attn_fwd_kernel, one block per work item); bit 0 = attn_fwd_kernel always; bit 1 = attn_fwd_ps_kernel
                                 * always (packed / windowed batches: its work items claimed from a counter); bit 2 = every step of
                                 * attn_bwd_dkdv4_kernel through its C++ body instead of the generated asm loops (bit-identical; A/B
                                 * and parity tests); bit 3 = with bit 1, packed / windowed batches keep the static deal of the items
                                 * (bit-identical to the claimed deal; measured: profiles/r06zv_attn_packed_ab.jsonl). (Rounds 2-3 kept three more opt-in kernels behind this knob -- a 4-wave x 64-row forward, the
                                 * round-1 dK/dV kernel, a 4-wave dQ kernel -- all measured at parity or slower: removed in round 4,
                                 * git 4501bb3:tools/experiments/attention_removed_r04.hip) */
#define UAMD_TUNE_RMS_VAR 5     /* RMSNorm kernels: 0 = one wave per row (row in registers, shuffle reduction),
                                 * 1 = one 256-thread block per row (one LDS reduction, 8 blocks per CU, several passes) */
#define UAMD_TUNE_GEMM_HALF 6   /* uamd_gemm_nt_256 tile height: 1 = 128-row tiles when the 256-row tiling has
                                 * fewer than 192 tiles (default), 0 = always 256 rows, 2 = always 128 rows */
#define UAMD_TUNE_GEMM_PERSIST 7 /* uamd_gemm_n{t,n}_256 with 256-row tiles: one persistent block per CU walks the tiles
                                 * and prefetches the next tile's first K tiles during the current one's last: 1 = when every CU gets
                                 * >= 4 tiles (default), 2 = whenever it gets more than one, 0 = never (one block per tile) */
#define UAMD_TUNE_DEQUANT_X4 8  /* row-major NF4 dequant to a 16-bit dtype: 1 = four 8-element groups per lane per
                                 * trip, loads issued ahead, shift instead of the 64-bit division (default), 0 = one group per lane */
#define UAMD_TUNE_GEMM_PLAIN 9  /* persistent 256x256 GEMM without accumulate / bias: 1 = the kernel instance whose
                                 * epilogue has no global loads (default: no vmcnt(0) in the K loop), 0 = the run-time-dispatch instance */
#define UAMD_TUNE_GLU_XA 10     /* (UAMD_GLU_XA) the gated activation fused with the LoRA rank products: 0 = 4 waves per 16-row block, two
                                 * 16-byte vectors per thread per tensor, tiles requested one step ahead (rounds 3-4); 1 = 8 waves, one
                                 * vector per thread; 2 = 8 waves and every tile requested two steps ahead; 3 (default) = 2 with the
                                 * columns of a row group split over adjacent workgroups (uamd_glu_{fwd,bwd}_xa_ws; without a
                                 * workspace: 2) where that measured faster -- rows of whole 4 KB pages, or at most 2048 rows --; 8 = always */
#define UAMD_TUNE_GEMM_S 11     /* (UAMD_GEMM_S) uamd_gemm_n{t,n}_256 on whole 256 x 256 tiles (M % 256 == 0, every N_g % 256 == 0, K >= 192):
                                 * 1 = the one-wave-per-SIMD kernel with the hand-ordered K loop (gemm_nt256s_kernel; default), as a
                                 * persistent walk when every CU gets at least four output tiles; 2 = the same, one workgroup per tile
                                 * always; 9 = the walk from two tiles per CU on; 0 = the 8-wave ping-pong kernels */
#define UAMD_TUNE_COUNT 12
int uamd_set_tuning(int knob, int value);
int uamd_gemm_nt_nf4(const void* A, int64_t lda, int M, int K, const uamd_gemm_group* groups,
                     int n_groups, int accumulate, int dtype, void* stream);
int uamd_lora_xa(const void* X, int64_t ldx, const void* A, int64_t lda, float* out,
                 int64_t ld_out, int M, int K, int R, int out_cols, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LoRA gradient products (fast_lora.py:172-189, :476-495, :632-637): up to 8 problems per launch,
 *     out = scale * P[M, R]^T @ Z[M, N]          (R <= 16; split wider ranks into 16-column problems)
 * P is fp32 (an uamd_lora_xa result) and is rounded to `dtype` on load, where the reference holds a tensor of
 * the activation dtype; Z is [M, N] in `dtype`; out is fp32, [R, N] (out_nr = 0, lora_A.grad layout) or
 * [N, R] (out_nr bit 0 set, lora_B.grad layout); out_nr bit 1 set: out += product (gradient accumulation
 * straight into the data-parallel arena) instead of out = product. Deterministic: fixed-order two-stage reduction over M through
 * `workspace` (fp32, >= sum_i ceil(M/128) * 16 * ceil(N_i/128)*128 floats always suffices). N % 8 == 0, ldz % 8 == 0, ldp % 4 == 0. */
typedef struct {
    const float* P;
    const void* Z;
    float* out;
    int64_t ldp, ldz, ldo;
    int N;
    int R;
    int out_nr;
    float scale;
} uamd_lora_tn_problem;
int uamd_lora_tn(const uamd_lora_tn_problem* probs, int n_probs, int M, float* workspace,
                 int64_t workspace_floats, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Causal GQA flash attention, head_dim D = a multiple of 8 up to 128 (csrc/attention.hip; below 128 the kernels keep their
 * 256-byte-row tiling and read nothing past a head's D elements as data). The step between RoPE and o_proj that the
 * reference delegates to flash-attn / xformers / SDPA (unsloth/utils/attention_dispatch.py:298-617, called from
 * unsloth/models/llama.py:757). Q [B,T,Hq,D], K/V [B,T,Hk,D], O [B,T,Hq,D] given by element strides
 * `strides` = {q_b,q_t,q_h, k_b,k_t,k_h, v_b,v_t,v_h, o_b,o_t,o_h} (d contiguous, multiples of 8);
 * LSE [B,Hq,lse_stride] fp32 (natural log-sum-exp of the scaled scores, saved for the backward; lse_stride =
 * T rounded up to a multiple of 32, pad zero-filled by the caller). Hq/Hk in 1 .. 8 (3, 5, 6, 7: a KV head's query heads run in
 * groups of 4 / 2 / 1 over the same K / V head -- no padded copies).
 * uamd_attn_bwd: two launches (dQ + Delta = rowsum(dO*O), then dK/dV), deterministic, no atomics. `strides` has
 * 24 entries: the 12 above, then dO, dQ, dK, dV (b, t, h each). Delta is a [2,B,Hq,lse_stride] fp32 scratch (plane 0:
 * -rowsum(dO*O), the C operand of the second launch's dP MFMAs; plane 1: LSE*log2(e); both written by the first launch, read
 * by the second; one plane < 2^31 bytes).
 * Band (packed documents / sliding window; block-diagonal causal mask of utils/packing.py:650-693, window rule
 * `q - key < W`): query q attends keys lo[q] <= key <= q, equivalently key is seen by queries key <= q <= hi[key].
 * lo, hi: int32 [B, T], non-decreasing along T, lo[q] <= q <= hi[q]; NULL (both) = plain causal. Tiles outside
 * the band are skipped, not just masked. */
int uamd_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, const int64_t* strides,
                  int B, int T, int Hq, int Hk, int D, int lse_stride, float scale, int causal, const int* lo,
                  int dtype, void* stream);
int uamd_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                  void* dQ, void* dK, void* dV, float* Delta, const int64_t* strides, int B, int T, int Hq, int Hk,
                  int D, int lse_stride, float scale, int causal, const int* lo, const int* hi, int dtype,
                  void* stream);
/* NON-CAUSAL attention inside documents (round 4; the vision tower of BASELINE config 4: Qwen2-VL's ViT attends all patches of
 * an image / frame, `cu_seqlens` windows): uamd_attn_fwd_band with causal = 0 and uamd_attn_bwd with causal = 0 take the band as
 * DOCUMENT edges -- query q and key k attend each other iff lo[q] <= k <= hi[q], lo / hi = first / last position of q's document
 * (intervals: equivalently lo[k] <= q <= hi[k]); both arrays are required. With causal != 0 uamd_attn_fwd_band is uamd_attn_fwd
 * (hi ignored). Same kernels: the upper edge of a query is a per-lane value instead of the query's own position. */
int uamd_attn_fwd_band(const void* Q, const void* K, const void* V, void* O, float* LSE, const int64_t* strides,
                       int B, int T, int Hq, int Hk, int D, int lse_stride, float scale, int causal, const int* lo,
                       const int* hi, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Single-token decode: csrc/decode.hip.
 *
 * uamd_gemv: y_g[n] = W_g[n, :] . x (+ lora_scale_g * LB_g[n, :] . t_g + bias_g[n]) for up to 4 row groups sharing one
 * token's x [K] (q|k|v, gate|up) in ONE launch. nf4 != 0: W_g is bitsandbytes-format NF4 (packed [N, K/2]; absmax per
 * `blocksize` codes: fp32, or nested uint8 codes + code2 + absmax2 + offset decoded in the same kernel). Replaces
 * fast_gemv and the bsz == 1 branch of fast_linear_forward (unsloth/kernels/utils.py:872-977, :1082-1125: bitsandbytes'
 * cdequantize_blockwise_fp32 + cgemm_4bit_inference_naive_{fp16,bf16}, then torch mv / addmv for LoRA). t_g = A_g x comes
 * from a previous uamd_gemv over the A rows with y_f32 = 1. K % 8 == 0 (nf4: K % 32 == 0, blocksize % 32 == 0),
 * K <= 16384, R <= 64. */
typedef struct {
    const void* W;             /* nf4: uint8 [N, K/2]; else dtype [N, K], row stride ldw elements */
    const uint8_t* absmax_u8;  /* nf4, nested: codes [N*K/blocksize] (with code2, absmax2, blocksize2, offset) */
    const float* absmax_f32;   /* nf4, single level: [N*K/blocksize]; takes precedence */
    const float* code2;        /* 256-entry map of the nested level */
    const float* absmax2;
    void* y;                   /* dtype [N] (float [N] when y_f32) */
    const float* lora_t;       /* fp32 [R] = A x, or NULL */
    const void* lora_b;        /* [N, ld_lb]: fp32 when lora_b_f32, else dtype */
    const void* bias;          /* dtype [N] or NULL */
    int64_t ldw, ld_lb;
    float offset, lora_scale;
    int N, R, blocksize2, lora_b_f32, y_f32, _pad;
} uamd_gemv_group;
int uamd_gemv(const void* x, int K, const uamd_gemv_group* groups, int n_groups, int nf4, int blocksize, int dtype,
              void* stream);
/* uamd_gemv_fused: the same launch with the token PRODUCED inside it and the LoRA `t = A x` computed INSIDE it, once, so that
 * one decoder layer of a decode step is 5 launches (q|k|v, attention, o, gate|up -> h, down) instead of 14
 * (LlamaModel_fast_forward_inference, llama.py:1249-1364: residual adds, fast_rms_layernorm_inference, fast_swiglu_inference
 * and the `mv` of fast_linear_forward are separate torch / kernel launches there).
 *   mode 0: x as given.   mode 1: x = (x * sigmoid(x)).to(T) * x2 (SwiGLU; x = gate, x2 = up).
 *   mode 2: h = T(x + res) (x may be NULL: h = res), x' = rmsnorm(h; eps) * norm_w (norm_w in T, or fp32 when w_f32);
 *           h is also written to h_out when non-NULL (by one block; h_out must not alias res or x).
 *   a_rows: [Rt, K] stacked LoRA A rows in T (row stride ld_a), Rt <= 256; group g's t starts at row t_off[g]; a group
 *           takes part when its lora_b / R / lora_scale are set (its lora_t is ignored). NULL: lora_t as in uamd_gemv.
 *           The first workgroups of the launch compute t (a row, or a 4096-column part of one, per wave) and publish each
 *           value as an 8-byte {value, tag} granule in `sync`; the weight-row workgroups pick the granules up after their
 *           own dot products (producers never wait, so the hand-off cannot deadlock whatever the residency; polls are bounded).
 *   sync:   DEVICE workspace of UAMD_GEMV_SYNC_BYTES, zeroed ONCE by the caller, required with a_rows; launches that share it
 *           must be ordered (one stream / one graph).
 *   tag, tag_dev: the launch's tag = tag + UAMD_TAG_STRIDE * *tag_dev (tag_dev NULL: tag alone), never 0 and never a value an
 *           earlier launch on the same workspace used: a host-side counter for eager launches; for launches replayed from a
 *           hipGraph a per-launch-site constant < UAMD_TAG_STRIDE plus a device counter the caller advances once per replay.
 *   glu:    2 groups (gate, up) of the same N: a wave computes row n of both and stores ONE value
 *           h[n] = (e * sigmoid(e)).to(T) * g at groups[0].y (e, g rounded to T first: the values the separate launches
 *           would have stored; fast_swiglu_inference, llama.py:572-606). K <= 8192. */
#define UAMD_GEMV_SYNC_BYTES (8 * 256)
#define UAMD_TAG_STRIDE 1024u
typedef struct {
    int mode, Rt, w_f32, glu;
    const void* x2;
    const void* res;
    const void* norm_w;
    void* h_out;
    const void* a_rows;
    int64_t ld_a;
    float eps;
    int t_off[4];
    unsigned tag;
    void* sync;
    const int* tag_dev;
} uamd_gemv_prologue;
int uamd_gemv_fused(const void* x, int K, const uamd_gemv_group* groups, int n_groups, int nf4, int blocksize, int dtype,
                    void* stream, const uamd_gemv_prologue* pro);
/* RoPE (rotate-half; the training kernel's arithmetic and rounding points) on the new token's q and k in
 * place in the fused row qkv [B, (Hq + 2 Hk) D], and append of k, v to the cache [B, Hk, s_max, D] at position
 * kv_len[b] (a DEVICE array: the step is replayable as a hipGraph). rope_pos (device, NULL = kv_len) indexes the
 * cos / sin tables [positions, >= D/2]. Replaces the six in-place torch ops + two permuted copies of
 * LlamaAttention_fast_forward_inference (unsloth/models/llama.py:468-497). */
int uamd_rope_kv_append(void* qkv, int64_t ld_qkv, const void* cos_t, const void* sin_t, int64_t ld_cs,
                        const int* kv_len, const int* rope_pos, void* k_cache, void* v_cache, int64_t cache_sb,
                        int64_t cache_sh, int B, int Hq, int Hk, int D, int s_max, int dtype, void* stream);
/* Split-KV decode attention over the cache, D = 128, GQA by head index: out[b, h, :] = softmax(q[b, h] . K^T * scale) V
 * over keys [max(0, len - window), len), len = kv_len[b] + len_add. Grid (nsplit, Hk, B), split s owns keys
 * [s * split_keys, (s + 1) * split_keys) (split_keys % 16 == 0, nsplit * split_keys >= s_max); partials = fp32
 * workspace [B, Hq, nsplit, D + 2]; a second launch combines them. Replaces llama.py:499-543 (expand + matmul +
 * softmax + matmul over the whole cache, or SDPA). */
int uamd_attn_decode(const void* q, int64_t q_sb, const void* k_cache, const void* v_cache, int64_t cache_sb,
                     int64_t cache_sh, const int* kv_len, int len_add, float* partials, void* out, int64_t out_sb,
                     int B, int Hq, int Hk, int D, int nsplit, int split_keys, int window, float scale, int dtype,
                     void* stream);
/* The three launches above (RoPE + append, split attention, combine) as ONE: every workgroup (split, kv head, batch) rotates
 * the G query heads it needs from the raw q|k|v row (qkv is NOT modified); the workgroup whose split owns position kv_len[b]
 * rotates the new k, appends k and v to the cache and uses them from LDS. Keys [max(0, len - window), len), len = kv_len[b] + 1.
 * The combine happens inside the launch, in split order (bit-identical to uamd_attn_decode's):
 *   nsplit * Hk * B <= 256 workgroups (all resident at once) and a launch tag given (tag / tag_dev as in uamd_gemv_prologue):
 *   partials travel as 8-byte {value, tag} granules and every workgroup combines its 1 / nsplit of the outputs; otherwise
 *   the last workgroup of a (batch, kv head) to arrive combines (a release / acquire fence pair and an arrival counter).
 * partials: workspace of B * Hq * nsplit * (D + 2) * 8 bytes, zeroed ONCE by the caller. counters: int32 [B * Hk], zeroed ONCE
 * by the caller (reset by the kernel). One stream of launches per workspace. */
int uamd_attn_decode_fused(const void* qkv, int64_t ld_qkv, const void* cos_t, const void* sin_t, int64_t ld_cs,
                           const int* kv_len, const int* rope_pos, void* k_cache, void* v_cache, int64_t cache_sb,
                           int64_t cache_sh, float* partials, int* counters, void* out, int64_t out_sb, int B, int Hq,
                           int Hk, int D, int s_max, int nsplit, int split_keys, int window, float scale, unsigned tag,
                           const int* tag_dev, int dtype, void* stream);

/* Greedy next token of a decode step: out[r] = argmax_i x[r, i] over `rows` contiguous fp32 rows of n logits (smallest index on
 * ties, like `logits.argmax(-1)` in HF's generate, llama.py:2167-2259 -> transformers). Two launches, 8 us for 128,256 logits
 * against 46 us for torch's reduction. Workspaces: ws_val rows * 64 floats, ws_idx rows * 64 int64. */
int uamd_argmax_f32(const float* x, int rows, int64_t n, float* ws_val, int64_t* ws_idx, int64_t* out, void* stream);

/* uamd_lora_xa2: same contract as uamd_lora_xa for R <= 64, streaming version (csrc/lora_side.hip): 32 rows per
 * block, K split over 4 waves, X and W through a per-wave LDS-DMA ring, fixed-order reduction. */
int uamd_lora_xa2(const void* X, int64_t ldx, const void* A, int64_t lda, float* out,
                  int64_t ld_out, int M, int K, int R, int out_cols, int dtype, void* stream);
/* uamd_lora_xa2k: uamd_lora_xa2 that ALSO writes out_k[M, k_cols] = T(X A^T) in the activation dtype (columns
 * >= R zero): the rank block uamd_gemm_nt_256 consumes as extra K tiles (uamd_gemm_group.lora_xk). k_cols % 8 == 0,
 * k_cols >= R, ld_k % 8 == 0. out (fp32) may be NULL when only the rank block is wanted. */
int uamd_lora_xa2k(const void* X, int64_t ldx, const void* A, int64_t lda, float* out, int64_t ld_out,
                   void* out_k, int64_t ld_k, int k_cols, int M, int K, int R, int out_cols, int dtype,
                   void* stream);

/* Activation-dtype copies of fp32 LoRA factors, row-major and transposed, for ALL matrices in one launch (the
 * reference casts per use: utils.py:1166-1167, fast_lora.py:138-145). `descs_dev` / `tile_prefix_dev` are DEVICE
 * arrays: n_mats descriptors and the exclusive prefix sum of ceil(rows/32)*ceil(cols/32) tiles per matrix;
 * total_tiles = their sum. dst_* may be NULL. */
typedef struct {
    const void* src;          /* fp32 [rows, cols] row-major */
    void* dst_rowmajor;       /* dtype [rows, cols] or NULL */
    void* dst_transposed;     /* dtype [cols, rows] or NULL */
    int rows, cols;
    /* optional third copy: pad_scale * src (pad_transposed == 0: element (r, c) at dst_pad[r * pad_ld + c]) or its
     * transpose (pad_transposed != 0: at dst_pad[c * pad_ld + r]) -- the zero-padded BK operand of
     * uamd_gemm_group.lora_bk (the caller zero-fills the buffer once; only the rank columns are written) */
    void* dst_pad;
    int64_t pad_ld;
    float pad_scale;
    int pad_transposed;
} uamd_lora_prep_desc;
int uamd_lora_prepare(const uamd_lora_prep_desc* descs_dev, const int* tile_prefix_dev, int n_mats,
                      int total_tiles, int dtype, void* stream);

/* AdamW (torch.optim.AdamW arithmetic: decoupled weight decay, bias-corrected moments, fp32) over ONE flat arena of n
 * parameters p with gradients g and moments m, v in the same order: one launch per optimizer step instead of the 52
 * multi_tensor_apply launches torch's fused AdamW needs for the 448 LoRA factors (the step that closes the reference's
 * training step, unsloth/trainer.py:445-623 via HF Trainer). bias_correction1 = 1 - beta1^t, bias_correction2_sqrt =
 * sqrt(1 - beta2^t) for step t >= 1; grad_scale multiplies g first (clipping; 1 = none); zero_grad != 0 writes zeros
 * back into g in the same pass. Scalars are doubles (derived constants such as 1 - beta2 are formed in double and
 * rounded once). All four pointers 16-byte aligned. */
int uamd_adamw_flat(float* p, float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, double bias_correction1, double bias_correction2_sqrt, double grad_scale,
                    int zero_grad, void* stream);
/* uamd_adamw_shard: the same arithmetic for FULL fine-tuning (BASELINE config 3; reference: loader.py:487-523 hands
 * full_finetuning to HF Trainer's optimizer over bf16 parameters). One rank's shard of a flat parameter bucket: fp32
 * master copy p32 and moments m, v (updated in place), the reduce-scattered gradient shard g16 in `dtype` (bf16 / fp16),
 * and the updated parameters rounded ONCE to `dtype` into p16 -- the slice of the bucket the all-gather broadcasts.
 * p32 / m / v 16-byte aligned, g16 / p16 8-byte aligned. */
int uamd_adamw_shard(float* p32, const void* g16, void* p16, float* m, float* v, int64_t n, double lr, double beta1,
                     double beta2, double eps, double weight_decay, double bias_correction1,
                     double bias_correction2_sqrt, double grad_scale, int dtype, void* stream);

/* debug: (lane,reg) -> (row,col) map of v_mfma_f32_16x16x32_bf16; out = float[2][64][4] */
int uamd_debug_mfma_probe(float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNSLOTH_AMD_H */
