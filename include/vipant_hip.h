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

const char* vipant_last_error(void);
int32_t vipant_version(void);
/* 0 when the current device is gfx950 and the code object loads; VIPANT_EHIP otherwise. */
int32_t vipant_device_check(void);

/* ---- dense contractions (torch.nn.Linear / F.conv2d-as-GEMM / nn.MultiheadAttention projections:
 *      cvap/module/val.py:500-506, 245-247, 288-289) ------------------------------------------------
 * C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous, bf16; fp32 accumulate on MFMA).
 * K % 64 == 0, N % 4 == 0.  bias (fp32 [N]) may be NULL.  `aux` is R (EPI_RESIDUAL_F32, may alias C),
 * U out (EPI_QUICKGELU) or U in (EPI_DQUICKGELU).  ldc applies to C and aux. */
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
/* out fp32 [M, D] = x fp32 + add bf16 (the last block's residual add, no norm behind it). */
int32_t vipant_residual_add(const float* x, const uint16_t* add, float* out, int64_t n, void* stream);
/* dx_f32[M,D] = dres (optional fp32 residual-stream gradient, may alias dx) + LN'(dy); dx_bf16 optional.
 * dy is bf16 [M,D] when dy_is_f32 == 0, fp32 otherwise.  dgamma / dbeta fp32 [D] (+)= column reductions;
 * dx_colsum (optional fp32 [D]) (+)= sum over rows of the produced dx: the bias gradient of the Linear whose
 * output gradient this dx is (out_proj / c_proj), for free in the same pass. */
size_t vipant_layernorm_bwd_workspace_bytes(int64_t M, int64_t D);
int32_t vipant_layernorm_bwd(const void* dy, int32_t dy_is_f32, const float* x, int64_t ldx, const float* mean,
                             const float* rstd, const float* gamma, const float* dres, float* dx_f32, int64_t lddx,
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

/* ---- elementwise / layout helpers ------------------------------------------------------------------
 * fp32 -> bf16 cast of a [R, C] matrix; dst_t (optional) receives the transpose [C, R]. */
int32_t vipant_cast_bf16(const float* src, uint16_t* dst, uint16_t* dst_t, int64_t R, int64_t C, void* stream);
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
int32_t vipant_scatter_rows(const float* g, const int64_t* idx, float* dx, int64_t n, int64_t rows_per_item,
                            int64_t D, void* stream);

/* ---- InfoNCE (CELossHead.forward, cvap/module/decoder/loss_head.py:265-284) ---------------------------
 * x1, x2 fp32 [B, E] (L2-normalised); logit_scale is the raw parameter (s = min(exp(.), scale_max),
 * scale_max <= 0 means +inf).  loss = mean_i CE(s x1 x2^T, i) + mean_i CE(s x2 x1^T, i).
 * Gradients are produced for rows [row0, row0+nrows) only (the rank's slice of an all-gathered batch):
 * dx1, dx2 fp32 [nrows, E], scaled by grad_scale; dlogit_scale is the full-batch value * grad_scale.
 * Any of dx1 / dx2 / dlogit_scale may be NULL (forward only).  E % 64 == 0. */
size_t vipant_infonce_workspace_bytes(int64_t B, int64_t E);
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

#ifdef __cplusplus
}
#endif
#endif /* VIPANT_HIP_H */
