/* vipant_hip.h -- C ABI of libvipant_hip.so, the MI355X (gfx950) hot path of the VIP-ANT
 * bimodal contrastive training step.
 *
 * Conventions (SURVEY.md section 8b, row B-c):
 *   - every entry point returns int32_t: 0 = ok, negative = VIPANT_E*; vipant_last_error()
 *     returns a thread-local message for the last failure;
 *   - arguments are raw DEVICE pointers + explicit dims + a hipStream_t passed as void*;
 *   - no allocation, no ownership transfer, no implicit synchronisation: the caller (PyTorch)
 *     owns every buffer, including workspaces whose size the *_workspace_bytes() queries return;
 *   - bf16 tensors are passed as uint16_t* (raw bfloat16 bits), row-major, last dim contiguous;
 *   - token-major activations are [M, D] with M = batch * tokens (batch-first row order
 *     m = b * S + s; the reference's seq-first [S, b, D] is the same math on permuted storage,
 *     cvap/module/encoder/clip_head.py:108-110).
 *
 * Each function cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef VIPANT_HIP_H
#define VIPANT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIPANT_OK 0
#define VIPANT_EBADSHAPE (-1)
#define VIPANT_EBADDTYPE (-2)
#define VIPANT_EALIGN (-3)
#define VIPANT_EHIP (-4)
#define VIPANT_ENOWORKSPACE (-5)

/* GEMM epilogues (vipant_gemm_nt). */
#define VIPANT_EPI_BF16 0          /* C_bf16 = acc (+bias) */
#define VIPANT_EPI_F32 1           /* C_f32  = acc (+bias) */
#define VIPANT_EPI_RESIDUAL_F32 2  /* C_f32  = acc + bias + R_f32   (out_proj / c_proj + residual) */
#define VIPANT_EPI_QUICKGELU 3     /* U_bf16 = acc + bias ; C_bf16 = U * sigmoid(1.702 U)  (c_fc) */
#define VIPANT_EPI_DQUICKGELU 4    /* C_bf16 = acc * dQuickGELU(U_bf16)        (backward of c_fc act) */
#define VIPANT_EPI_SCALE_F32 5     /* C_f32  = alpha * acc */
#define VIPANT_EPI_QUICKGELU_D8 6  /* as QUICKGELU, but aux receives uint8 codes of QuickGELU'(U) instead of U (1 B / element) */
#define VIPANT_EPI_DQUICKGELU_D8 7 /* C_bf16 = acc * decode(aux uint8): backward of c_fc act from the 8-bit derivative code */
/* OR-ed into `epilogue`: the rows of this launch are ONE ROW PER ITEM of a batch (the last block on its read-out rows, the read-out
 * projection).  Such launches take 64 x 64 tiles with K split over the waves of a workgroup instead of 256 x 256 tiles (21-91 us ->
 * 9-30 us at 512 rows).  A property of the call site, not of M: a row's result must not depend on how many rows travel with it
 * (`running.micro_batch` reproduces the full-batch loss to 1e-6), so the kernel choice may not either.  Epilogues: BF16, F32,
 * RESIDUAL_F32, QUICKGELU_D8, DQUICKGELU_D8. */
#define VIPANT_EPI_FEW_ROWS 0x100

const char* vipant_last_error(void);
int32_t vipant_version(void);
/* 0 when the current device is gfx950 and the code object loads; VIPANT_EHIP otherwise. */
int32_t vipant_device_check(void);

/* ---- dense contractions (torch.nn.Linear / F.conv2d-as-GEMM / nn.MultiheadAttention projections:
 *      cvap/module/val.py:500-506, 245-247, 288-289) ------------------------------------------------
 * C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous, bf16; fp32 accumulate on MFMA).
 * K % 64 == 0, N % 4 == 0.  bias (fp32 [N]) may be NULL.  `aux` is R (EPI_RESIDUAL_F32, may alias C),
 * U out (EPI_QUICKGELU) or U in (EPI_DQUICKGELU), or the uint8 [M, N] matrix of QuickGELU' codes (the _D8 epilogues: the backward
 * needs only the derivative, code = round((QuickGELU'(U) + 0.1) * 212.5), error <= 2.4e-3).  ldc applies to C and aux (elements). */
int32_t vipant_gemm_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                       const float* bias, void* aux, float alpha, int64_t M, int64_t N, int64_t K,
                       int32_t epilogue, void* stream);

/* Weight-gradient contraction (autograd of nn.Linear weight): C[P,Q] (+)= A[M,P]^T . B[M,Q], reduction
 * over the token dimension M (both operands M-major).  Q % 4 == 0 (P arbitrary).  fp32 output;
 * accumulate != 0 adds into C.  Deterministic split over M through `workspace`
 * (vipant_gemm_tn_workspace_bytes).  a_colsum (optional fp32 [P]) (+)= sum_m A[m, p]: the bias gradient of the
 * same Linear, taken from the A tiles while they sit in LDS (no extra pass over dY). */
size_t vipant_gemm_tn_workspace_bytes(int64_t M, int64_t P, int64_t Q);
int32_t vipant_gemm_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                       int64_t M, int64_t P, int64_t Q, int32_t accumulate, float* a_colsum, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Two contractions of one shape in one launch: C0 = A0^T . B0 and C1 = A1^T . B1 (the two feature gradients of the InfoNCE loss,
 * cvap/module/decoder/loss_head.py:276-283: together they fill the chip with half the splits of two separate launches). */
size_t vipant_gemm_tn_pair_workspace_bytes(int64_t M, int64_t P, int64_t Q);
int32_t vipant_gemm_tn_pair(const uint16_t* A0, const uint16_t* B0, float* C0, const uint16_t* A1, const uint16_t* B1, float* C1,
                            int64_t lda, int64_t ldb, int64_t ldc, int64_t M, int64_t P, int64_t Q, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Column sums over tokens (bias gradients): out[N] (+)= sum_m X[m, n]; X bf16 [M, N]. */
size_t vipant_colsum_workspace_bytes(int64_t M, int64_t N);
int32_t vipant_colsum_bf16(const uint16_t* X, int64_t ldx, float* out, int64_t M, int64_t N, int32_t accumulate,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm (clip/model.py:154-160: fp32 statistics, eps 1e-5) -----------------------------------
 * x fp32 rows (row stride ldx elements) -> y bf16 [M, D] (optional); mean / rstd fp32 [M] saved for backward.
 * y_f32 (optional, may be NULL) receives the fp32 result as well (ln_pre writes the residual stream).
 * add (optional bf16 [M, D]): the residual add `x + branch` of ResidualAttentionBlock.forward (cvap/module/val.py:520-521)
 * fused in front of the norm; sum_out (optional fp32 [M, D]) receives x + add (the new residual stream). */
int32_t vipant_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, uint16_t* y,
                             float* y_f32, float* mean, float* rstd, int64_t M, int64_t D, const uint16_t* add,
                             float* sum_out, void* stream);
/* Precision of the residual stream INSIDE the transformer stack (`stream_flags` of the operators below): fp32, or fp16 -- the
 * reference's own autocast precision (clip/model.py:157-160 casts the fp32 LayerNorm result back to the fp16 stream) -- with fp32
 * statistics and the norm taken on the unrounded sum.  IN: the `x` argument is fp16; OUT: the `x_out` / `sum_out` argument is. */
#define VIPANT_STREAM_IN_F16 1
#define VIPANT_STREAM_OUT_F16 2
#define VIPANT_STREAM_FEW_ROWS 0x100 /* (vipant_ln_mlp_quickgelu_bwd_e4m3) its contractions are VIPANT_EPI_FEW_ROWS launches */
#define VIPANT_STREAM_ACT_Q 0x200 /* (vipant_ln_qkv_bwd_e4m3) the plan's activation scratch already holds dqkv's e4m3 form (vipant_mha_bwd_e4m3) */
/* out fp32 [M, D] = x (fp32, or fp16 with VIPANT_STREAM_IN_F16) + add bf16 (the last block's residual add, no norm behind it). */
int32_t vipant_residual_add(const void* x, const uint16_t* add, float* out, int64_t n, int32_t stream_flags, void* stream);
/* dx[M,D] = dres (optional residual-stream gradient) + LN'(dy); outputs dx_f32 (optional) and dx_bf16 (optional).
 * flags: VIPANT_LN_DY_F32 -- dy is fp32 [M,D] (else bf16); VIPANT_LN_DRES_BF16 -- dres is bf16 [M,D] (row stride D; may be the
 * same buffer as dx_bf16), else fp32 with row stride lddx (may alias dx_f32).  dgamma / dbeta fp32 [D] (+)= column reductions;
 * dx_colsum (optional fp32 [D]) (+)= sum over rows of the produced dx: the bias gradient of the Linear whose
 * output gradient this dx is (out_proj / c_proj), for free in the same pass. */
size_t vipant_layernorm_bwd_workspace_bytes(int64_t M, int64_t D);
#define VIPANT_LN_DY_F32 1
#define VIPANT_LN_DRES_BF16 2
#define VIPANT_LN_X_F16 4 /* x rows are fp16 (a saved fp16 stream) */
int32_t vipant_layernorm_bwd(const void* dy, int32_t flags, const void* x, int64_t ldx, const float* mean,
                             const float* rstd, const float* gamma, const void* dres, float* dx_f32, int64_t lddx,
                             uint16_t* dx_bf16, float* dgamma, float* dbeta, float* dx_colsum, int32_t accumulate,
                             int64_t M, int64_t D, void* workspace, size_t workspace_bytes, void* stream);

/* ---- multi-head attention core (nn.MultiheadAttention inside ResidualAttentionBlock,
 *      cvap/module/val.py:511-517): softmax(q k^T / sqrt(64) [+ causal mask]) v, head dim 64 -----------
 * qkv bf16 [batch*S, 3*D] packed q|k|v (in_proj row order), out bf16 [batch*S, D]; lse fp32 [batch, H, S]
 * (natural-log row log-sum-exp of the scaled scores, saved for backward). */
int32_t vipant_mha_fwd(const uint16_t* qkv, uint16_t* out, float* lse, int64_t batch, int64_t S, int64_t H,
                       int32_t causal, void* stream);
/* dqkv bf16 [batch*S, 3*D] from dout bf16 [batch*S, D]; delta fp32 [batch, H, S] is scratch. */
int32_t vipant_mha_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse,
                       float* delta, uint16_t* dqkv, int64_t batch, int64_t S, int64_t H, int32_t causal,
                       void* stream);

/* The patch embedding written straight into the token matrix (ViTPreEncoder, cvap/module/val.py:249-257): for m = item * P + patch,
 * tokens[item * (P + 1) + patch + 1, :] = A[m, :] . B^T + pos[patch + 1, :] (fp32, ldc = N); the class-token rows
 * tokens[item * S, :] = cls + pos[0, :] are vipant_tokens_cls_rows'. */
int32_t vipant_gemm_nt_tokens(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* tokens, const float* pos,
                              int64_t b, int64_t P, int64_t N, int64_t K, void* stream);
int32_t vipant_tokens_cls_rows(const float* cls, const float* pos, float* tokens, int64_t b, int64_t S, int64_t D, void* stream);

/* ---- the same attention for ONE query per (item, head): the last block of a tower on its read-out rows ----------
 * Both read-outs take one token per item from the stack's output (the class token, cvap/module/val.py:288-289; the
 * end-of-text token, val.py:143-145), so only that row of the last block's output is ever read and only it carries a
 * gradient: the block's per-token work runs on `batch` rows, and its attention has one query per (item, head) against all
 * keys (exact: dead rows eliminated, nothing approximated).  Row of item i: i * S + idx[i] (idx == NULL: the first row).
 * q_rows bf16 [batch, D]: the projected queries of those rows; qkv bf16 [batch*S, 3*D]: only its K and V column blocks are
 * read; causal: keys <= the row.  out_rows bf16 [batch, D]; probs fp32 [batch, H, S]: the softmax rows, saved for backward. */
int32_t vipant_mha_rows_fwd(const uint16_t* q_rows, const uint16_t* qkv, const int64_t* idx, uint16_t* out_rows, float* probs,
                            int64_t batch, int64_t S, int64_t H, int32_t causal, void* stream);
/* dq_rows bf16 [batch, D]; dK / dV of EVERY key row written into the K / V column blocks of dqkv bf16 [batch*S, 3*D] (zero
 * rows behind a causal limit); the Q column block of dqkv is not touched. */
int32_t vipant_mha_rows_bwd(const uint16_t* q_rows, const uint16_t* qkv, const int64_t* idx, const float* probs,
                            const uint16_t* dout_rows, uint16_t* dq_rows, uint16_t* dqkv, int64_t batch, int64_t S, int64_t H,
                            int32_t causal, void* stream);
/* The same one-query attention with the K / V projection folded into the query side (csrc/readout_ctx.hip): with one query per
 * (item, head) the scores are (W_k,h^T q_h) . h1_j / 8 (+ a constant the softmax drops) and the head's output is
 * W_v,h (sum_j p_j h1_j) + b_v,h, so the kernels run against the LayerNorm output h1 bf16 [batch*S, D] itself and the key / value
 * projection of every token (nn.MultiheadAttention's in_proj, clip/model.py:170-187 via cvap/module/val.py:519-522) is never formed.
 * qk bf16 [batch*H, D]: row (i, h) = W_k,h^T q_(i,h);  ctx bf16 [batch*H, D]: row (i, h) = sum_j p_j h1[i, j, :];
 * probs fp32 [batch, H, S] as above.  H = 8, 12 or 16 heads of 64 (D = 64 H); S <= 1024 (an item's raw scores wait
 * in LDS beside the staged rows: 64 KiB at H = 16).
 * pair != 0 (round 5, the step's default): qk, ctx (and dqk below; not dctx) are bf16 PAIRS, x = hi + lo, stored as two planes
 * [2][batch*H, D] (hi plane first): these per-(item, head) vectors carry 16 instead of 8 mantissa bits between the kernels that
 * produce and consume them, so the folded form adds no rounding the reference's attention does not have. */
int32_t vipant_rows_ctx_fwd(const uint16_t* qk, const uint16_t* h1, const int64_t* idx, uint16_t* ctx, float* probs, int64_t batch,
                            int64_t S, int64_t H, int32_t causal, int32_t pair, void* stream);
/* dctx bf16 [batch*H, D] (a single plane also with pair != 0): row (i, h) = W_v,h^T dout_(i,h).  dh1 bf16 [batch*S, D]: the attention's
 * gradient for EVERY token's h1 row (zeros behind a causal limit; all rows written);  dqk bf16 [batch*H, D] (a pair with pair != 0):
 * gradient of qk;  dbk fp32 [D] or NULL: the key bias's gradient, exact zeros (a bias on the keys cannot move a softmax over keys). */
int32_t vipant_rows_ctx_bwd(const uint16_t* qk, const uint16_t* dctx, const uint16_t* ctx, const uint16_t* h1, const int64_t* idx,
                            const float* probs, uint16_t* dh1, uint16_t* dqk, float* dbk, int64_t batch, int64_t S, int64_t H,
                            int32_t causal, int32_t pair, void* stream);
/* H independent products in one launch, C_h[M, N] = A_h[M, K] . B_h[N, K]^T (+ bias_h), bf16 in and out, operand h starting
 * h * stride elements behind operand 0: the per-head contractions of the folded form (a head's 64 columns of the `batch` read-out
 * rows against that head's block of W_k / W_v, and back) without the block-sparse [batch * H, D] operand of vipant_head_expand.
 * a_lo / c_lo != 0: A / C is a bf16 pair (see vipant_rows_ctx_fwd) whose lo plane lies a_lo / c_lo elements behind the hi plane: a
 * pair in A contributes both planes to the product, a pair in C receives bf16(acc) and bf16(acc - hi). */
int32_t vipant_gemm_nt_heads(const uint16_t* A, int64_t lda, int64_t stride_a, int64_t a_lo, const uint16_t* B, int64_t ldb,
                             int64_t stride_b, uint16_t* C, int64_t ldc, int64_t stride_c, int64_t c_lo, const float* bias,
                             int64_t stride_bias, int64_t M, int64_t N, int64_t K, int64_t H, void* stream);
/* rows bf16 [n, 64 H] -> out bf16 [n*H, 64 H]: row (i, h) = row i with every column outside head h's 64 zeroed -- the operand that
 * makes "per-head slice times the head's weight block" one dense contraction.  The step uses it for the two weight gradients of
 * the folded form (dW_v = expand(do)^T ctx, dW_k = expand(q)^T dqk through vipant_gemm_tn); the activations' products go through
 * vipant_gemm_nt_heads. */
int32_t vipant_head_expand(const uint16_t* rows, uint16_t* out, int64_t n, int64_t H, void* stream);
/* its inverse on a product: rows[i, 64 h + c] = full[(i, h), 64 h + c] (+ bias[64 h + c]); full bf16 or fp32 [n*H, 64 H]. */
int32_t vipant_head_extract(const void* full, int32_t full_is_f32, const float* bias, uint16_t* rows, int64_t n, int64_t H,
                            void* stream);
/* dst row i <- src row (i * rows_per_item + idx[i]) (idx == NULL: + 0); rows of row_bytes bytes, a multiple of 16. */
int32_t vipant_gather_rows_bytes(const void* src, const int64_t* idx, void* dst, int64_t n, int64_t rows_per_item, int64_t row_bytes,
                           void* stream);
/* x bf16 [*, D]: row (i * rows_per_item + idx[i]) <- bf16(row + add row i); add compact [n, D], fp32 (add_is_f32) or bf16. */
int32_t vipant_add_rows_bf16(uint16_t* x, const int64_t* idx, const void* add, int32_t add_is_f32, int64_t n,
                             int64_t rows_per_item, int64_t D, void* stream);

/* ---- elementwise / layout helpers ------------------------------------------------------------------
 * fp32 -> bf16 cast of a [R, C] matrix; dst_t (optional) receives the transpose [C, R]. */
int32_t vipant_cast_bf16(const float* src, uint16_t* dst, uint16_t* dst_t, int64_t R, int64_t C, void* stream);
/* The same for a list of matrices in one launch (all weight matrices of a tower once per step): device arrays of ntensors device
 * pointers src / dst / dst_t (entries of dst or dst_t may be NULL), row / column counts R, C, and tile_start[ntensors + 1], the
 * running sum of ceil(R/64) * ceil(C/64); total_tiles = tile_start[ntensors]. */
int32_t vipant_cast_bf16_multi(const float* const* src, uint16_t* const* dst, uint16_t* const* dst_t, const int32_t* R,
                               const int32_t* C, const int32_t* tile_start, int64_t ntensors, int64_t total_tiles, void* stream);
/* bf16 -> fp32 widening of n elements (n % 4 == 0). */
int32_t vipant_cast_f32(const uint16_t* src, float* dst, int64_t n, void* stream);
/* fp8 operands (BASELINE.json configs[4], "fp8 MFMA weights"; the reference has no counterpart -- its contractions are the
 * fp16 autocast matmuls of clip/model.py:170-187 -- so parity is against an exact emulation of this quantiser, tests/test_fp8_gpu.py).
 * bf16 rows x [M, K] -> OCP e4m3 rows q [M, K] with one power-of-two scale per row: scale[m] = e + 127 where e is the smallest
 * exponent with max|x[m, :]| / 2^e <= 448, q = round-to-nearest-even(x / 2^e).  K % 8 == 0, K <= 8192. */
int32_t vipant_quant_e4m3_rows(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                               void* stream);
/* The block format for ACTIVATIONS (round 5; what v_mfma_scale_f32_16x16x128_f8f6f4's scale operands take): one power-of-two scale per
 * 32 consecutive elements of a row, e = the smallest exponent with max|block| / 2^e <= 448, byte = e + 127, so that a producer can
 * quantise the 32 columns it holds without knowing the rest of the row.  The scale bytes are stored in the order the contraction
 * consumes them: byte of (row m, block kb = k / 32) at (((m / 128) * (K / 128) + kb / 4) * 16 + m % 16) * 32 + (kb % 4) * 8 + (m % 128) / 16;
 * vipant_mx_scale_bytes(M, K) = ceil(M / 128) * (K / 128) * 512 bytes (0 if K % 128 != 0).  K % 128 == 0. */
size_t vipant_mx_scale_bytes(int64_t M, int64_t K);
int32_t vipant_quant_e4m3_mx(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                             void* stream);
/* The same for K columns (K % 32 == 0) that start at column 32 * kb0 of rows 128 * kt_row elements long in the scale layout; x and q
 * point at the first of those columns. */
int32_t vipant_quant_e4m3_mx_cols(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                                  int64_t kt_row, int64_t kb0, void* stream);
/* Block-UNIFORM scales (round 6): the same bytes-plus-scales format with every scale shared by an aligned block of 32 rows x 32
 * columns (the byte is stored for each of the 32 rows, in the layout above, so the NT contractions read the matrix unchanged).  One
 * scale per (32 tokens, 32 columns) is also one scale per 32 k of a COLUMN, which is what v_mfma_scale_f32_16x16x128_f8f6f4 needs when k
 * runs along the token axis: the operand format of vipant_gemm_tn_e4m3.  vipant_quant_e4m3_mx32: from bf16 (e = the smallest exponent
 * with max|block| / 2^e <= 448 over the 32 x 32 block).  vipant_mx_uniform32: IN PLACE on an e4m3 matrix with row-wise block scales (what
 * the producers emit): the block's scale becomes the largest of its rows' scales, the other rows' bytes are divided by the power of
 * two -- exact unless a value falls below e4m3's normal range.  No reference counterpart (BASELINE.json configs[4]). */
int32_t vipant_quant_e4m3_mx32(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                               void* stream);
int32_t vipant_mx_uniform32(uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K, void* stream);
/* The same two for K columns (K % 32 == 0) that start at column 32 * kb0 of rows 128 * kt_row elements long in the scale layout; x and q
 * point at the first of those columns (as vipant_quant_e4m3_mx_cols). */
int32_t vipant_quant_e4m3_mx32_cols(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                                    int64_t kt_row, int64_t kb0, void* stream);
int32_t vipant_mx_uniform32_cols(uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K, int64_t kt_row, int64_t kb0, void* stream);
/* Weight-gradient contraction on e4m3 operands (BASELINE.json configs[4]; the autograd of nn.Linear weight, cvap/module/val.py:500-506,
 * as vipant_gemm_tn): C[P, Q] fp32 (+)= dequant(A, sa)^T dequant(B, sb), reduction over the token dimension M.  A [M, P], B [M, Q]:
 * token-major e4m3 bytes with block-uniform scales (above); lda / ldb = the row length of the quantised matrices (what their scale
 * layouts were written with), P and Q multiples of 128.  fp32 accumulation, deterministic split over M through `workspace`.
 * a_colsum (optional fp32 [P]) (+)= sum_m dequant(A)[m, p], taken from the A tiles while they sit in LDS, as in vipant_gemm_tn. */
size_t vipant_gemm_tn_e4m3_workspace_bytes(int64_t M, int64_t P, int64_t Q);
int32_t vipant_gemm_tn_e4m3(const uint8_t* A, int64_t lda, const uint8_t* sa, const uint8_t* B, int64_t ldb, const uint8_t* sb, float* C,
                            int64_t ldc, int64_t M, int64_t P, int64_t Q, int32_t accumulate, float* a_colsum, void* workspace,
                            size_t workspace_bytes, void* stream);
/* vipant_mha_fwd / vipant_mha_bwd that also leave the e4m3 + block-scale form of their result for the contraction that follows
 * (out_proj; in_proj^T): `out` [M, D] and the dQ columns of `dqkv` [M, 3 D] -- made by a pass behind the kernel -- in the block-uniform
 * form (vipant_quant_e4m3_mx32, bit for bit; round 6), the dK | dV columns the streamed backward emits from its own epilogue in the
 * row-wise form (vipant_quant_e4m3_mx, bit for bit): vipant_mx_uniform32_cols makes those uniform too where a weight gradient wants them.  No reference counterpart
 * (BASELINE.json configs[4]).  The streamed single-pass backward (224 < S <= 320, no mask) emits the dK | dV columns from its own
 * epilogue; everything else -- dQ, the other backward shapes, the forward -- is the stand-alone pass enqueued behind the kernel.
 * H even (D % 128 == 0). */
int32_t vipant_mha_fwd_e4m3(const uint16_t* qkv, uint16_t* out, float* lse, uint8_t* out_q, uint8_t* out_scale, int64_t batch,
                            int64_t S, int64_t H, int32_t causal, void* stream);
int32_t vipant_mha_bwd_e4m3(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta,
                            uint16_t* dqkv, uint8_t* dqkv_q, uint8_t* dqkv_scale, int64_t batch, int64_t S, int64_t H,
                            int32_t causal, void* stream);
/* C[M, N] (bf16) = dequant(A, sa) dequant(B, sb)^T (+ bias) on v_mfma_scale_f32_16x16x128_f8f6f4 (fp32 accumulation; the scales ride
 * the instruction's block-scale operands).  A [M, K] e4m3: an activation, sa = its BLOCK scales (vipant_quant_e4m3_mx layout); B [N, K]
 * e4m3: a weight matrix, sb = one scale per row (vipant_quant_e4m3_rows).  Epilogues: VIPANT_EPI_BF16, VIPANT_EPI_QUICKGELU_D8,
 * VIPANT_EPI_DQUICKGELU_D8, meaning as in vipant_gemm_nt.  cq / cq_scale (both or neither; the two QuickGELU epilogues, N % 128 == 0):
 * the epilogue ALSO leaves the e4m3 form of its bf16 result, bytes [M, N] + block scales -- vipant_quant_e4m3_mx of C bit for bit --
 * i.e. the next contraction's A operand quantised where it is produced; with cq set and VIPANT_EPI_QUICKGELU_D8, C and aux may be
 * NULL (only the e4m3 form is wanted).  K % 128 == 0, K >= 256, N % 8 == 0, leading dimensions (in elements = bytes) multiples of 16. */
int32_t vipant_gemm_nt_e4m3(const uint8_t* A, int64_t lda, const uint8_t* sa, const uint8_t* B, int64_t ldb, const uint8_t* sb,
                            void* C, int64_t ldc, const float* bias, void* aux, uint8_t* cq, uint8_t* cq_scale, int64_t M, int64_t N,
                            int64_t K, int32_t epilogue, void* stream);
/* conv1.weight [O, Cin, kh, kw] fp32 -> effective GEMM weight bf16 [O, Cout*kh*kw]; mean_channels != 0
 * averages the Cin stored channels into one (cvap/module/val.py:236-244), else Cout = Cin. */
int32_t vipant_conv_weight_prep(const float* w, uint16_t* out, int64_t O, int64_t Cin, int64_t khw,
                                int32_t mean_channels, void* stream);
/* im2col for the patch conv (cvap/module/val.py:245-252): x fp32 [b, C, T, F] -> bf16 [b*nrow*ncol, C*ph*pw],
 * patch (i, j) at (i*sh, j*sw), row order t-major (token = i*ncol + j). */
int32_t vipant_im2col(const float* x, uint16_t* out, int64_t b, int64_t C, int64_t T, int64_t F, int64_t ph,
                      int64_t pw, int64_t sh, int64_t sw, void* stream);
/* tokens[b, 0, :] = cls + pos[0]; tokens[b, 1+p, :] = patches[b*P + p, :] + pos[1+p]  (val.py:253-257). fp32. */
int32_t vipant_assemble_tokens(const float* patches, const float* cls, const float* pos, float* tokens, int64_t b,
                               int64_t P, int64_t D, void* stream);
/* backward of assemble_tokens: dpatches bf16 [b*P, D], dcls [D] and dpos [S, D] fp32 (+)=. */
int32_t vipant_assemble_tokens_bwd(const float* dtokens, uint16_t* dpatches, float* dcls, float* dpos,
                                   int32_t accumulate, int64_t b, int64_t P, int64_t D, void* stream);
/* effective-kernel gradient [O, khw] fp32 -> conv1.weight.grad [O, Cin, khw] (+)= g / Cin per channel. */
int32_t vipant_conv_weight_grad(const float* g, float* wgrad, int64_t O, int64_t Cin, int64_t khw,
                                int32_t accumulate, void* stream);
/* y = x / ||x||_2 per row (clip_head.py:117-118); norm fp32 [M] saved. fp32 [M, E]. */
int32_t vipant_l2norm_fwd(const float* x, float* y, float* norm, int64_t M, int64_t E, void* stream);
/* dx = (dy - y * <dy, y>) / norm; dx_bf16 optional copy. */
int32_t vipant_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, uint16_t* dx_bf16,
                          int64_t M, int64_t E, void* stream);
/* token_embedding[text] + pos[:L]  (GPTPreEncoder, val.py:117-121): tokens i64 [b, L] -> x fp32 [b*L, D];
 * eot i64 [b] = argmax over the row. */
int32_t vipant_embed_tokens(const int64_t* tokens, const float* table, const float* pos, float* x, int64_t* eot,
                            int64_t b, int64_t L, int64_t D, void* stream);
/* gather rows: out[i, :] = x[i * rows_per_item + idx[i], :] (EOT read-out, val.py:145); idx NULL = row 0. fp32. */
int32_t vipant_gather_rows(const float* x, const int64_t* idx, float* out, int64_t n, int64_t rows_per_item,
                           int64_t D, void* stream);
/* scatter-add of the read-out gradient back into a zeroed token-major gradient (backward of gather_rows). */
int32_t vipant_scatter_rows_bf16(const float* g, const int64_t* idx, uint16_t* dx_bf16, uint16_t* compact_bf16, int64_t n,
                                 int64_t rows_per_item, int64_t D, void* stream);
int32_t vipant_scatter_rows(const float* g, const int64_t* idx, float* dx, int64_t n, int64_t rows_per_item,
                            int64_t D, void* stream);

/* ---- InfoNCE (CELossHead.forward, cvap/module/decoder/loss_head.py:265-284) ---------------------------
 * x1, x2 fp32 [B, E] (L2-normalised); logit_scale is the raw parameter (s = min(exp(.), scale_max),
 * scale_max <= 0 means +inf).  loss = mean_i CE(s x1 x2^T, i) + mean_i CE(s x2 x1^T, i).
 * Gradients are produced for rows [row0, row0+nrows) only (the rank's slice of an all-gathered batch):
 * dx1, dx2 fp32 [nrows, E], scaled by grad_scale; dlogit_scale is the full-batch value * grad_scale.
 * Any of dx1 / dx2 / dlogit_scale may be NULL (forward only).  E % 64 == 0.  Up to 768 clips at E = 512 the call is three
 * launches of row-block kernels (no B x B operand in HBM but the fp32 logits themselves, 2 B^2 floats); above, 256 x 256 tile
 * kernels whose s.dZ makes a bf16 round trip.  Same results within the rounding of the bf16 gradient operands.
 * Strip form (round 5; B > 768, 0 < nrows <= 768, E = 512 -- one rank of an N-GPU step): pass 1 keeps the fp32-grade logits of the
 * strip's rows of Z and of Z^T (2 nrows B floats), one launch forms the strip's gradients from them; nothing B x B is stored, and
 * vipant_infonce_strip_workspace_bytes(B, E, nrows) (55 MB at B = 4096 / 512 rows against 181 MB) is enough workspace. */
size_t vipant_infonce_workspace_bytes(int64_t B, int64_t E);
size_t vipant_infonce_strip_workspace_bytes(int64_t B, int64_t E, int64_t nrows);
int32_t vipant_infonce_fwd_bwd(const float* x1, const float* x2, const float* logit_scale, float scale_max,
                               float* loss, float* dx1, float* dx2, float* dlogit_scale, float grad_scale,
                               int64_t B, int64_t E, int64_t row0, int64_t nrows, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- Retrieval evaluation (LossHead.report / retrieval_eval, cvap/module/decoder/loss_head.py:71-168) ----
 * x1 fp32 [N1, E] queries, x2 fp32 [N2, E] candidates (both L2-normalised by the caller), gold int32 [N1, G]
 * with entries in [0, N2).  Replaces `(x1 @ x2.t()).argsort(descending=True)` + `torch.where(ind == label)`:
 *   ranks[i, g] = #{ j : sim[i, j] > sim[i, gold[i, g]] }   (0-based position of the gold column in the sort)
 *   top1[i]     = argmax_j sim[i, j] (lowest j on ties); may be NULL.
 * The N1 x N2 similarity matrix is never stored.  E % 64 == 0, 1 <= G <= 64. */
size_t vipant_retrieval_workspace_bytes(int64_t N1, int64_t N2, int64_t E, int64_t G);
int32_t vipant_retrieval_ranks(const float* x1, const float* x2, const int32_t* gold, int32_t* ranks, int32_t* top1,
                               int64_t N1, int64_t N2, int64_t E, int64_t G, void* workspace,
                               size_t workspace_bytes, void* stream);

/* ---- Log-mel front-end (cvap/data/audio/transform.py:12-35 -> torchaudio.compliance.kaldi.fbank with the parameters
 * of cvap/data/image_audio.py:119-126; padding / normalisation / SpecAugment masks of image_audio.py:183-207) ----
 * wave fp32 [b, wave_stride] (clip i valid for nsamples[i] samples), out fp32 [b, T, F] (= [b, 1, T, F]).
 * Frames: snip_edges, window_size / window_shift samples, zero-padded to nfft (512 / 1024 / 2048); per-frame DC removal,
 * pre-emphasis, `window` [window_size] (symmetric Hann for the reference), power spectrum, `banks` fp32 [F, nfft/2+1]
 * triangular mel weights whose non-zero support of row m is [bank_start[m], bank_start[m] + bank_len[m]), log with the
 * fp32-epsilon floor.  Frames past a clip's last whole window are 0 before normalisation.  zero_mean != 0 subtracts the
 * clip mean first (zero_mean_wf).  norm_std != 0: (x - norm_mean) / norm_std.  masks int32 [b, 4] = f0, f1, t0, t1
 * (SpecAugment: bins [f0, f1) and frames [t0, t1) set to 0), or NULL. */
size_t vipant_fbank_workspace_bytes(int64_t b);
int32_t vipant_fbank(const float* wave, int64_t wave_stride, const int64_t* nsamples, float* out, const float* window,
                     const float* banks, const int32_t* bank_start, const int32_t* bank_len, const int32_t* masks,
                     int64_t b, int64_t T, int64_t F, int32_t window_size, int32_t window_shift, int32_t nfft,
                     float preemphasis, int32_t zero_mean, float norm_mean, float norm_std, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ---- LARS (cvap/module/lars.py:43-72), one fused pass per tensor list ---------------------------------
 * For tensor i (n[i] elements): dp = g + wd*p (adapt[i]); q = eta*|p|/|dp| (adapt[i], both norms > 0);
 * mu = momentum*mu + q*dp; p -= lr[i]*mu.  ptrs are device arrays of device pointers. */
size_t vipant_lars_workspace_bytes(int64_t ntensors);
int32_t vipant_lars_step(float* const* p, const float* const* g, float* const* mu, const int64_t* n,
                         const int32_t* adapt, const float* lr, int64_t ntensors, float weight_decay,
                         float momentum, float eta, void* workspace, size_t workspace_bytes, void* stream);

/* ==== The fused operator set: one entry point per reference operator group, forward and backward ==========
 * Host-side compositions of the kernels above (vipant_amd/csrc/block.hip): same stream, no allocation, no synchronisation.
 * Fusion plan: a residual add rides on the NEXT LayerNorm pass (`add` bf16 [M,D] = the previous branch output, `x_out` fp32
 * [M,D] = x + add, both NULL for the first block); QuickGELU / QuickGELU' are epilogues of the c_fc / c_proj^T contractions;
 * bias gradients are column sums taken inside the weight-gradient contraction or inside the LayerNorm backward (`dx_colsum`).
 * Weights are bf16 copies of the fp32 parameters: `w` as stored ([out, in]), `w_t` transposed ([in, out]).
 * `workspace` of the backward entry points: vipant_block_workspace_bytes(M, D) bytes, 256-byte aligned. */
size_t vipant_block_workspace_bytes(int64_t M, int64_t D);

/* K2 -- ln_1 + packed in_proj of nn.MultiheadAttention (cvap/module/val.py:519-520, 500; clip/model.py:154-160).
 * h bf16 [M,D] = LN(x (+ add)), mean / rstd fp32 [M], qkv bf16 [M,3D] = h . w_qkv^T + b_qkv. */
int32_t vipant_ln_qkv_fwd(const float* x, const uint16_t* add, float* x_out, const float* gamma, const float* beta,
                          const uint16_t* w_qkv, const float* b_qkv, uint16_t* h, float* mean, float* rstd, uint16_t* qkv,
                          int64_t M, int64_t D, void* stream);
/* dstream fp32 [M,D]: in = gradient of the residual stream after this block's attention branch, out = gradient before the block
 * (in place); dx_bf16 its bf16 copy.  dstream == NULL: the gradient stream is kept in bf16 only -- dx_bf16 is read as the incoming
 * gradient and overwritten with the outgoing one (the forward stream stays fp32; profiles/r2_stream_precision.md, model D).
 * dh bf16 [M,D] scratch; dw fp32 [3D,D], db fp32 [3D], dgamma / dbeta fp32 [D];
 * dx_colsum (optional fp32 [D]) = column sums of the produced gradient (= d c_proj.bias of the block below). */
int32_t vipant_ln_qkv_bwd(const uint16_t* dqkv, const uint16_t* w_qkv_t, const uint16_t* h, const float* x, const float* mean,
                          const float* rstd, const float* gamma, float* dstream, uint16_t* dx_bf16, uint16_t* dh, float* dw,
                          float* db, float* dgamma, float* dbeta, float* dx_colsum, int64_t M, int64_t D, void* workspace,
                          size_t workspace_bytes, void* stream);

/* K4 -- out_proj (cvap/module/val.py:517, 520).  residual == NULL: out bf16 [M,N] = a . w^T + bias (the step's form: the add
 * happens in the next LayerNorm pass); residual fp32 [M,N]: out fp32 [M,N] = a . w^T + bias + residual (stand-alone form). */
int32_t vipant_gemm_bias_residual_fwd(const uint16_t* a, const uint16_t* w, const float* bias, const float* residual, void* out,
                                      int64_t M, int64_t N, int64_t K, void* stream);
/* da bf16 [M,K] = dy . w (w_t = w^T, [K,N]); dw fp32 [N,K] = dy^T a.  (d bias = column sums of dy: see dx_colsum above.) */
int32_t vipant_gemm_bias_residual_bwd(const uint16_t* dy, const uint16_t* w_t, const uint16_t* a, uint16_t* da, float* dw,
                                      int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes, void* stream);

/* K5 -- ln_2 + c_fc + QuickGELU + c_proj (cvap/module/val.py:502-506, 521; clip/model.py:163-165).
 * h bf16 [M,D] = LN(x (+ add)); u = h . w_fc^T + b_fc; g bf16 [M,4D] = u * sigmoid(1.702 u); dcode uint8 [M,4D] = 8-bit code of
 * QuickGELU'(u) (all the backward needs of u: VIPANT_EPI_QUICKGELU_D8); y bf16 [M,D] = g . w_proj^T + b_proj. */
int32_t vipant_ln_mlp_quickgelu_fwd(const float* x, const uint16_t* add, float* x_out, const float* gamma, const float* beta,
                                    const uint16_t* w_fc, const float* b_fc, const uint16_t* w_proj, const float* b_proj,
                                    uint16_t* h, float* mean, float* rstd, uint8_t* dcode, uint16_t* g, uint16_t* y, int64_t M,
                                    int64_t D, void* stream);
/* dcode, g again from the saved h (activation-memory plan `running.recompute_mlp`). */
int32_t vipant_mlp_quickgelu_recompute(const uint16_t* h, const uint16_t* w_fc, const float* b_fc, uint8_t* dcode, uint16_t* g,
                                       int64_t M, int64_t D, void* stream);
/* dy bf16 [M,D] = gradient of the MLP branch output (= bf16 copy of the stream gradient); dstream as in vipant_ln_qkv_bwd;
 * du bf16 [M,4D], dh bf16 [M,D] scratch; dw_proj fp32 [D,4D], dw_fc fp32 [4D,D], db_fc fp32 [4D];
 * dx_colsum (optional fp32 [D]) = d out_proj.bias. */
int32_t vipant_ln_mlp_quickgelu_bwd(const uint16_t* dy, const uint16_t* w_proj_t, const uint16_t* w_fc_t, const uint8_t* dcode,
                                    const uint16_t* g, const uint16_t* h, const float* x, const float* mean, const float* rstd,
                                    const float* gamma, float* dstream, uint16_t* dx_bf16, uint16_t* du, uint16_t* dh,
                                    float* dw_proj, float* dw_fc, float* db_fc, float* dgamma, float* dbeta, float* dx_colsum,
                                    int64_t M, int64_t D, void* workspace, size_t workspace_bytes, void* stream);

/* ---- The same block operators with e4m3 operands in their NT contractions (BASELINE.json configs[4], "fp8 MFMA weights") ----
 * `plan` carries the pre-quantised weights (vipant_quant_e4m3_rows of the SAME matrices, in the SAME orientation, as the bf16
 * weight arguments, which are then unused and may be NULL) and the scratch the operator quantises its activation into; every NT
 * contraction becomes vipant_gemm_nt_e4m3 on an activation block-quantised by its producer (a LayerNorm pass, the epilogue of the
 * contraction before it) or, for the two attention outputs, by vipant_quant_e4m3_mx.  The weight-gradient contractions, LayerNorm, the
 * attention core and the residual stream are unchanged (bf16 / fp32).  plan == NULL: exactly the bf16 operator.
 * w_q / w_scale: the operator's (first) weight; w2_q / w2_scale: the second weight of the MLP operators (forward: w_fc then w_proj;
 * backward: w_proj_t then w_fc_t), one scale per row; act_q: bytes [M, 4D] (the widest activation); act_scale: its block scales, one
 * per 32 elements of a row in the MX layout of vipant_quant_e4m3_mx, vipant_mx_scale_bytes(M, 4D) bytes. */
typedef struct vipant_fp8_plan {
    const uint8_t* w_q;
    const uint8_t* w_scale;
    const uint8_t* w2_q;
    const uint8_t* w2_scale;
    uint8_t* act_q;
    uint8_t* act_scale;
    /* optional (both or neither), bytes [M, 4D] + vipant_mx_scale_bytes(M, 4D): where the epilogue of an MLP operator's first
     * contraction (c_fc + QuickGELU forward, c_proj^T * QuickGELU' backward) leaves the e4m3 form of its result for the second one
     * (round 5: the producer quantises, no pass over the [M, 4D] activation).  NULL: the result is block-quantised by a separate pass. */
    uint8_t* emit_q;
    uint8_t* emit_scale;
    /* backward operators, optional (both or neither): bytes [M, D] + vipant_mx_scale_bytes(M, D) that accompany the bf16 stream gradient `dx_bf16` between
     * operators.  On entry they hold the quantised form of the incoming gradient (the operator's first contraction then skips its
     * quantisation pass); an operator that produces a new stream gradient (its LayerNorm backward) writes the new one's there. */
    uint8_t* dy_q;
    uint8_t* dy_scale;
    /* round 6, backward operators: tn_e4m3 != 0 -- the operator's WEIGHT-GRADIENT contractions run on e4m3 operands too
     * (vipant_gemm_tn_e4m3; emit_q / emit_scale are then required as a second scratch).  Operands whose e4m3 form the operator has at
     * hand (dy_q; du in emit_q; dqkv in act_q) are made block-uniform in place (vipant_mx_uniform32); the kept forward activation
     * -- g (MLP), the attention output (out_proj), ln_1's output (in_proj) -- is read from keep_q / keep_scale, ln_2's output from
     * keep2_q / keep2_scale, if the forward kept their block-uniform e4m3 forms, and is otherwise quantised from the bf16 argument
     * (vipant_quant_e4m3_mx32) into scratch.  The bias gradients (in_proj, c_fc) ride on the contraction as in the bf16 operators. */
    int64_t tn_e4m3;
    const uint8_t* keep_q;
    const uint8_t* keep_scale;
    const uint8_t* keep2_q;
    const uint8_t* keep2_scale;
} vipant_fp8_plan;
/* LayerNorm with the block quantisation of its bf16 output fused (the row is in registers anyway): q bytes [M, D] / qscale bytes
 * [vipant_mx_scale_bytes(M, D)] = vipant_quant_e4m3_mx of y resp. dx_bf16, bit for bit; both NULL: exactly vipant_layernorm_fwd /
 * vipant_layernorm_bwd. */
int32_t vipant_layernorm_fwd_e4m3(const void* x, int64_t ldx, const float* gamma, const float* beta, uint16_t* y, float* y_f32,
                                  float* mean, float* rstd, int64_t M, int64_t D, const uint16_t* add, void* sum_out, uint8_t* q,
                                  uint8_t* qscale, int32_t stream_flags, void* stream);
int32_t vipant_layernorm_bwd_e4m3(const void* dy, int32_t flags, const void* x, int64_t ldx, const float* mean, const float* rstd,
                                  const float* gamma, const void* dres, float* dx_f32, int64_t lddx, uint16_t* dx_bf16,
                                  float* dgamma, float* dbeta, float* dx_colsum, int32_t accumulate, int64_t M, int64_t D,
                                  void* workspace, size_t workspace_bytes, uint8_t* q, uint8_t* qscale, void* stream);
int32_t vipant_ln_qkv_fwd_e4m3(const void* x, const uint16_t* add, void* x_out, const float* gamma, const float* beta,
                               const uint16_t* w_qkv, const float* b_qkv, uint16_t* h, float* mean, float* rstd, uint16_t* qkv,
                               int64_t M, int64_t D, const vipant_fp8_plan* plan, int32_t stream_flags, void* stream);
int32_t vipant_ln_qkv_bwd_e4m3(const uint16_t* dqkv, const uint16_t* w_qkv_t, const uint16_t* h, const void* x, const float* mean,
                               const float* rstd, const float* gamma, float* dstream, uint16_t* dx_bf16, uint16_t* dh, float* dw,
                               float* db, float* dgamma, float* dbeta, float* dx_colsum, int64_t M, int64_t D, void* workspace,
                               size_t workspace_bytes, const vipant_fp8_plan* plan, int32_t stream_flags, void* stream);
int32_t vipant_gemm_bias_residual_fwd_e4m3(const uint16_t* a, const uint16_t* w, const float* bias, const float* residual, void* out,
                                           int64_t M, int64_t N, int64_t K, const vipant_fp8_plan* plan, void* stream);
int32_t vipant_gemm_bias_residual_bwd_e4m3(const uint16_t* dy, const uint16_t* w_t, const uint16_t* a, uint16_t* da, float* dw,
                                           int64_t M, int64_t N, int64_t K, void* workspace, size_t workspace_bytes,
                                           const vipant_fp8_plan* plan, void* stream);
int32_t vipant_ln_mlp_quickgelu_fwd_e4m3(const void* x, const uint16_t* add, void* x_out, const float* gamma, const float* beta,
                                         const uint16_t* w_fc, const float* b_fc, const uint16_t* w_proj, const float* b_proj,
                                         uint16_t* h, float* mean, float* rstd, uint8_t* dcode, uint16_t* g, uint16_t* y, int64_t M,
                                         int64_t D, const vipant_fp8_plan* plan, int32_t stream_flags, void* stream);
int32_t vipant_mlp_quickgelu_recompute_e4m3(const uint16_t* h, const uint16_t* w_fc, const float* b_fc, uint8_t* dcode, uint16_t* g,
                                            int64_t M, int64_t D, const vipant_fp8_plan* plan, void* stream);
int32_t vipant_ln_mlp_quickgelu_bwd_e4m3(const uint16_t* dy, const uint16_t* w_proj_t, const uint16_t* w_fc_t, const uint8_t* dcode,
                                         const uint16_t* g, const uint16_t* h, const void* x, const float* mean, const float* rstd,
                                         const float* gamma, float* dstream, uint16_t* dx_bf16, uint16_t* du, uint16_t* dh,
                                         float* dw_proj, float* dw_fc, float* db_fc, float* dgamma, float* dbeta, float* dx_colsum,
                                         int64_t M, int64_t D, void* workspace, size_t workspace_bytes, const vipant_fp8_plan* plan,
                                         int32_t stream_flags, void* stream);

/* K1 -- ViTPreEncoder.forward (cvap/module/val.py:228-259): patch conv as im2col + contraction, cls token, positional table,
 * ln_pre.  x fp32 [b,C,T,F]; conv_w fp32 [Dw,Cw,ph,pw] (mean_channels != 0: the Cw stored channels are averaged, val.py:236-244);
 * scratch / saved: w_eff bf16 [Dw,kcols], patches bf16 [b*P,kcols], pe: unused since round 3 (may be NULL), tokens fp32 [b*S,Dw] (kcols =
 * (mean_channels ? 1 : Cw)*ph*pw, P = nrow*ncol, S = P+1); out fp32 [b*S,Dw] = the residual stream; mean / rstd fp32 [b*S]. */
int32_t vipant_patch_embed_ln_fwd(const float* x, const float* conv_w, const float* cls, const float* pos, const float* gamma,
                                  const float* beta, uint16_t* w_eff, uint16_t* patches, float* pe, float* tokens, float* out,
                                  float* mean, float* rstd, int64_t b, int64_t C, int64_t T, int64_t F, int64_t Dw, int64_t Cw,
                                  int64_t ph, int64_t pw, int64_t sh, int64_t sw, int32_t mean_channels, void* stream);
size_t vipant_patch_embed_ln_bwd_workspace_bytes(int64_t b, int64_t P, int64_t Dw, int64_t kcols);
/* dtokens fp32 [b*S,Dw], dpatches bf16 [b*P,Dw], dw_eff fp32 [Dw,kcols] scratch; dconv fp32 [Dw,Cw,khw] (mean_channels; else
 * dw_eff IS the weight gradient); dcls fp32 [Dw]; dpos fp32 [rows >= S, Dw] must arrive zeroed. */
/* dout: gradient of the residual stream the stack hands back -- fp32 [b*S,Dw], or (dout_bf16 != 0) its bf16 stream gradient as is. */
int32_t vipant_patch_embed_ln_bwd(const void* dout, int32_t dout_bf16, const float* tokens, const float* mean, const float* rstd, const float* gamma,
                                  const uint16_t* patches, float* dtokens, uint16_t* dpatches, float* dw_eff, float* dconv,
                                  float* dcls, float* dpos, float* dgamma, float* dbeta, int64_t b, int64_t P, int64_t Dw,
                                  int64_t Cw, int64_t khw, int32_t mean_channels, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* K6 -- ViTPostEncoder.forward + MetaHead's normalisation (cvap/module/val.py:288-289, clip_head.py:117-118):
 * feat fp32 [batch,E] = LN(x[b, idx_b]) . proj (proj_t = proj^T bf16 [E,D]); idx == NULL reads row 0 (cls) in place, else the
 * rows are gathered into `rows` fp32 [batch,D]; normalized != 0: out = feat / |feat|, norm fp32 [batch] (else out is unused). */
int32_t vipant_cls_ln_proj_l2norm_fwd(const float* x, const int64_t* idx, const float* gamma, const float* beta,
                                      const uint16_t* proj_t, float* rows, uint16_t* y, float* mean, float* rstd, float* feat,
                                      float* out, float* norm, int64_t batch, int64_t S, int64_t D, int64_t E, int32_t normalized,
                                      void* stream);
size_t vipant_cls_ln_proj_l2norm_bwd_workspace_bytes(int64_t batch, int64_t D, int64_t E);
/* dout fp32 [batch,E]; proj bf16 [D,E]; dfeat bf16 [batch,E], dy bf16 [batch,D], drows fp32 [batch,D] scratch; dx fp32 [batch*S,D]
 * zeroed by the caller; dproj fp32 [D,E]. */
/* dx (token-major, zeroed by the caller) receives the read-out rows' gradient; or, compact form: `drows` fp32 [batch, D] receives
 * them (always for idx != NULL; for the cls read-out when drows != NULL) and dx may be NULL -- the transformer stack's backward
 * scatters them into its bf16 stream gradient itself (vipant_scatter_rows_bf16) instead of reading a dense fp32 matrix of zeros. */
int32_t vipant_cls_ln_proj_l2norm_bwd(const float* dout, const float* out, const float* norm, const float* x, const int64_t* idx,
                                      const float* rows, const uint16_t* y, const float* mean, const float* rstd,
                                      const float* gamma, const uint16_t* proj, uint16_t* dfeat, uint16_t* dy, float* drows,
                                      float* dx, float* dproj, float* dgamma, float* dbeta, int64_t batch, int64_t S, int64_t D,
                                      int64_t E, int32_t normalized, void* workspace, size_t workspace_bytes, void* stream);

/* K7 -- GPTPreEncoder.forward (cvap/module/val.py:109-122) and GPTPostEncoder.forward (val.py:136-146) of the frozen text tower. */
int32_t vipant_embed_gather_pos_fwd(const int64_t* tokens, const float* table, const float* pos, float* x, int64_t* eot,
                                    int64_t b, int64_t L, int64_t D, void* stream);
int32_t vipant_eot_ln_proj_l2norm_fwd(const float* x, const int64_t* eot, const float* gamma, const float* beta,
                                      const uint16_t* proj_t, float* rows, uint16_t* y, float* mean, float* rstd, float* feat,
                                      float* out, float* norm, int64_t batch, int64_t L, int64_t D, int64_t E, int32_t normalized,
                                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VIPANT_HIP_H */
