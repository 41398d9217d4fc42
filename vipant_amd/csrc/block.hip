// The fused operator set of the C ABI (SURVEY.md 8b, row B-c): one entry point per reference operator group of the step --
// K1 patch embedding + ln_pre, K2 ln_1 + packed QKV projection, K4 out_proj, K5 ln_2 + c_fc + QuickGELU + c_proj, K6 / K7
// read-outs -- forward and backward each.  Every function here is host code only: it validates its arguments and enqueues the
// kernels of this library (gemm_nt.hip, gemm_tn.hip, layernorm.hip, elementwise.hip) on the caller's stream, in the order and
// with the fusion plan described in DESIGN.md section 5:
//   * a residual add is never an epilogue of the contraction that produces the branch -- it rides on the NEXT LayerNorm pass
//     (`add` / `x_out`), which reads and writes the fp32 stream anyway;
//   * QuickGELU and its derivative are epilogues of the c_fc / c_proj^T contractions;
//   * bias gradients come for free: column sums of dY inside the weight-gradient contraction (in_proj, c_fc) or inside the
//     LayerNorm backward that produces dY (out_proj, c_proj: `dx_colsum`).
// No allocation, no synchronisation; scratch activations and the split-reduction workspace are the caller's.
#include "common.h"

namespace {

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

#define TRY(expr)                      \
    do {                               \
        const int32_t _rc = (expr);    \
        if (_rc != VIPANT_OK) return _rc; \
    } while (0)

// a . w^T on the bf16 operands, or -- with a plan -- on their e4m3 forms: the weight comes pre-quantised (wq / ws: the plan's copy
// of the SAME matrix, same orientation, as the bf16 argument it replaces, one scale per row); the activation in MX block format
// (common.h): (aq, as) = already quantised by its producer (a LayerNorm pass, the epilogue of the contraction before), or NULL:
// block-quantised here into the plan's scratch.  (cq, cqs) != NULL: the epilogue also leaves the e4m3 form of the result there (MX
// scales) for the next contraction; `out` / `aux` may then be NULL (QuickGELU epilogue: only the e4m3 form is wanted).
int32_t nt(const vipant_fp8_plan* plan, const uint8_t* wq, const uint8_t* ws, const uint8_t* aq, const uint8_t* as,
           const uint16_t* a, const uint16_t* w, void* out, const float* bias, void* aux, int64_t M, int64_t N, int64_t K,
           int32_t epi, void* stream, uint8_t* cq = nullptr, uint8_t* cqs = nullptr) {
    if (plan == nullptr) return vipant_gemm_nt(a, K, w, K, out, N, bias, aux, 1.0f, M, N, K, epi, stream);
    VIPANT_REQUIRE(wq != nullptr && ws != nullptr && plan->act_q != nullptr && plan->act_scale != nullptr, VIPANT_EBADSHAPE,
                   "fp8 plan: quantised weight, its row scales and the activation scratch are all required");
    if (aq == nullptr) {
        TRY(vipant_quant_e4m3_mx(a, K, plan->act_q, K, plan->act_scale, M, K, stream));
        aq = plan->act_q; as = plan->act_scale;
    }
    return vipant_gemm_nt_e4m3(aq, K, as, wq, K, ws, out, N, bias, aux, cq, cqs, M, N, K, epi, stream);
}

// LayerNorm backward of a block operator: in place on the stream gradient (fp32 master + bf16 copy, or bf16 only), the new
// gradient's quantised form written beside it when the plan carries the buffers
int32_t ln_bwd(const vipant_fp8_plan* plan, const uint16_t* dh, const void* x, int32_t stream_flags, const float* mean,
               const float* rstd, const float* gamma, float* dstream, uint16_t* dx_bf16, float* dgamma, float* dbeta,
               float* dx_colsum, int64_t M, int64_t D, void* workspace, size_t workspace_bytes, void* stream) {
    uint8_t* q = plan ? plan->dy_q : nullptr;
    uint8_t* qs = plan ? plan->dy_scale : nullptr;
    const int32_t xf = (stream_flags & VIPANT_STREAM_IN_F16) ? VIPANT_LN_X_F16 : 0;       // the saved stream rows are fp16
    if (dstream == nullptr)
        return vipant_layernorm_bwd_e4m3(dh, VIPANT_LN_DRES_BF16 | xf, x, D, mean, rstd, gamma, dx_bf16, nullptr, D, dx_bf16, dgamma,
                                         dbeta, dx_colsum, 0, M, D, workspace, workspace_bytes, q, qs, stream);
    return vipant_layernorm_bwd_e4m3(dh, xf, x, D, mean, rstd, gamma, dstream, dstream, D, dx_bf16, dgamma, dbeta, dx_colsum, 0, M,
                                     D, workspace, workspace_bytes, q, qs, stream);
}

// Do this operator's weight-gradient contractions run on e4m3 operands?  (The plan asks for it, the second scratch is there, and the
// shapes are the kernel's: every operand width a multiple of 128.)
bool tn8(const vipant_fp8_plan* plan, int64_t D) {
    return plan != nullptr && plan->tn_e4m3 != 0 && plan->emit_q != nullptr && plan->emit_scale != nullptr && D % 128 == 0;
}

// One operand [M, K] of vipant_gemm_tn_e4m3 in block-uniform form: (q, s) given and row-wise -> made uniform in place; given and
// `uniform` -> as they are; not given -> quantised from the bf16 matrix into (scratch_q, scratch_s).
int32_t operand8(uint8_t* q, uint8_t* s, bool uniform, const uint16_t* bf16_src, uint8_t* scratch_q, uint8_t* scratch_s, int64_t M,
                 int64_t K, const uint8_t** out_q, const uint8_t** out_s, void* stream) {
    if (q != nullptr && s != nullptr) {
        if (!uniform) TRY(vipant_mx_uniform32(q, K, s, M, K, stream));
        *out_q = q; *out_s = s;
        return VIPANT_OK;
    }
    VIPANT_REQUIRE(bf16_src != nullptr, VIPANT_EBADSHAPE, "fp8 plan: a weight-gradient operand is neither kept in e4m3 nor given in bf16");
    TRY(vipant_quant_e4m3_mx32(bf16_src, K, scratch_q, K, scratch_s, M, K, stream));
    *out_q = scratch_q; *out_s = scratch_s;
    return VIPANT_OK;
}

}  // namespace

extern "C" size_t vipant_block_workspace_bytes(int64_t M, int64_t D) {
    size_t w = vipant_gemm_tn_workspace_bytes(M, 3 * D, D);
    w = max_sz(w, vipant_gemm_tn_workspace_bytes(M, 4 * D, D));
    w = max_sz(w, vipant_gemm_tn_workspace_bytes(M, D, 4 * D));
    w = max_sz(w, vipant_gemm_tn_workspace_bytes(M, D, D));
    return max_sz(w, vipant_layernorm_bwd_workspace_bytes(M, D));
}

// ------------------------------------------------------------------------------------------------ K2: ln_1 + in_proj
extern "C" int32_t vipant_ln_qkv_fwd_e4m3(const void* x, const uint16_t* add, void* x_out, const float* gamma, const float* beta,
                                          const uint16_t* w_qkv, const float* b_qkv, uint16_t* h, float* mean, float* rstd,
                                          uint16_t* qkv, int64_t M, int64_t D, const vipant_fp8_plan* plan, int32_t stream_flags,
                                          void* stream) {
    VIPANT_REQUIRE(M > 0 && D > 0 && D % 64 == 0, VIPANT_EBADSHAPE, "ln_qkv_fwd: bad shape M=%ld D=%ld", (long)M, (long)D);
    VIPANT_REQUIRE((add == nullptr) == (x_out == nullptr), VIPANT_EBADSHAPE, "ln_qkv_fwd: add and x_out go together");
    TRY(vipant_layernorm_fwd_e4m3(x, D, gamma, beta, h, nullptr, mean, rstd, M, D, add, x_out, plan ? plan->act_q : nullptr,
                                  plan ? plan->act_scale : nullptr, stream_flags, stream));
    return nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, plan ? plan->act_q : nullptr,
              plan ? plan->act_scale : nullptr, h, w_qkv, qkv, b_qkv, nullptr, M, 3 * D, D, VIPANT_EPI_BF16, stream);
}

extern "C" int32_t vipant_ln_qkv_fwd(const float* x, const uint16_t* add, float* x_out, const float* gamma, const float* beta,
                                     const uint16_t* w_qkv, const float* b_qkv, uint16_t* h, float* mean, float* rstd,
                                     uint16_t* qkv, int64_t M, int64_t D, void* stream) {
    return vipant_ln_qkv_fwd_e4m3(x, add, x_out, gamma, beta, w_qkv, b_qkv, h, mean, rstd, qkv, M, D, nullptr, 0, stream);
}

extern "C" int32_t vipant_ln_qkv_bwd_e4m3(const uint16_t* dqkv, const uint16_t* w_qkv_t, const uint16_t* h, const void* x,
                                          const float* mean, const float* rstd, const float* gamma, float* dstream,
                                          uint16_t* dx_bf16, uint16_t* dh, float* dw, float* db, float* dgamma, float* dbeta,
                                          float* dx_colsum, int64_t M, int64_t D, void* workspace, size_t workspace_bytes,
                                          const vipant_fp8_plan* plan, int32_t stream_flags, void* stream) {
    VIPANT_REQUIRE(M > 0 && D > 0 && D % 64 == 0, VIPANT_EBADSHAPE, "ln_qkv_bwd: bad shape M=%ld D=%ld", (long)M, (long)D);
    VIPANT_REQUIRE(workspace_bytes >= vipant_block_workspace_bytes(M, D), VIPANT_ENOWORKSPACE, "ln_qkv_bwd: workspace too small");
    // dh = dqkv . W_qkv  (NT on the transposed weight [D, 3D]); VIPANT_STREAM_ACT_Q: vipant_mha_bwd_e4m3 has left dqkv's e4m3 form
    // in the plan's scratch
    bool preq = plan != nullptr && (stream_flags & VIPANT_STREAM_ACT_Q);
    const bool wg8 = tn8(plan, D);
    if (wg8 && !preq) {      // no producer has quantised dqkv: do it here, block-uniform from the start (one pass serves both contractions)
        TRY(vipant_quant_e4m3_mx32(dqkv, 3 * D, plan->act_q, 3 * D, plan->act_scale, M, 3 * D, stream));
        preq = true;
    } else if (wg8) {
        // vipant_mha_bwd_e4m3's form: the dQ columns are block-uniform already (its pass makes them so); the dK | dV columns, where
        // the streamed kernel emitted them row-wise from its epilogue, are made uniform in place (a pass over their scale bytes alone
        // where they are uniform already)
        TRY(vipant_mx_uniform32_cols(plan->act_q + D, 3 * D, plan->act_scale, M, 2 * D, 3 * D / 128, D / 32, stream));
    }
    TRY(nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, preq ? plan->act_q : nullptr,
           preq ? plan->act_scale : nullptr, dqkv, w_qkv_t, dh, nullptr, nullptr, M, D, 3 * D, VIPANT_EPI_BF16, stream));
    // dW_qkv = dqkv^T h, d b_qkv = column sums of dqkv
    if (wg8) {
        const uint8_t *aq, *as, *bq, *bs;
        TRY(operand8(plan->act_q, plan->act_scale, true, dqkv, nullptr, nullptr, M, 3 * D, &aq, &as, stream));
        TRY(operand8(const_cast<uint8_t*>(plan->keep_q), const_cast<uint8_t*>(plan->keep_scale), true, h, plan->emit_q, plan->emit_scale,
                     M, D, &bq, &bs, stream));
        TRY(vipant_gemm_tn_e4m3(aq, 3 * D, as, bq, D, bs, dw, D, M, 3 * D, D, 0, db, workspace, workspace_bytes, stream));
    } else {
        TRY(vipant_gemm_tn(dqkv, 3 * D, h, D, dw, D, M, 3 * D, D, 0, db, workspace, workspace_bytes, stream));
    }
    // ln_1 backward + residual-gradient add, in place on the stream gradient (fp32 master + bf16 copy, or bf16 only)
    return ln_bwd(plan, dh, x, stream_flags, mean, rstd, gamma, dstream, dx_bf16, dgamma, dbeta, dx_colsum, M, D, workspace,
                  workspace_bytes, stream);
}

extern "C" int32_t vipant_ln_qkv_bwd(const uint16_t* dqkv, const uint16_t* w_qkv_t, const uint16_t* h, const float* x,
                                     const float* mean, const float* rstd, const float* gamma, float* dstream,
                                     uint16_t* dx_bf16, uint16_t* dh, float* dw, float* db, float* dgamma, float* dbeta,
                                     float* dx_colsum, int64_t M, int64_t D, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    return vipant_ln_qkv_bwd_e4m3(dqkv, w_qkv_t, h, x, mean, rstd, gamma, dstream, dx_bf16, dh, dw, db, dgamma, dbeta, dx_colsum, M,
                                  D, workspace, workspace_bytes, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------ K4: out_proj
extern "C" int32_t vipant_gemm_bias_residual_fwd_e4m3(const uint16_t* a, const uint16_t* w, const float* bias,
                                                      const float* residual, void* out, int64_t M, int64_t N, int64_t K,
                                                      const vipant_fp8_plan* plan, void* stream) {
    if (residual != nullptr) {      // stand-alone form: out fp32 = a . w^T + bias + residual (bf16 operands only)
        VIPANT_REQUIRE(plan == nullptr, VIPANT_EBADSHAPE, "gemm_bias_residual_fwd: the fp32-residual form has no e4m3 variant");
        return vipant_gemm_nt(a, K, w, K, out, N, bias, const_cast<float*>(residual), 1.0f, M, N, K, VIPANT_EPI_RESIDUAL_F32,
                              stream);
    }
    // the step's form: branch output as bf16; the add happens in the next LayerNorm pass
    // (a == NULL with a plan: the producer -- vipant_mha_fwd_e4m3 -- has left the operand's e4m3 form in the plan's scratch)
    VIPANT_REQUIRE(a != nullptr || plan != nullptr, VIPANT_EBADSHAPE, "gemm_bias_residual_fwd: no operand");
    const bool preq = plan != nullptr && a == nullptr;
    return nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, preq ? plan->act_q : nullptr,
              preq ? plan->act_scale : nullptr, a, w, out, bias, nullptr, M, N, K, VIPANT_EPI_BF16, stream);
}

extern "C" int32_t vipant_gemm_bias_residual_fwd(const uint16_t* a, const uint16_t* w, const float* bias,
                                                 const float* residual, void* out, int64_t M, int64_t N, int64_t K,
                                                 void* stream) {
    return vipant_gemm_bias_residual_fwd_e4m3(a, w, bias, residual, out, M, N, K, nullptr, stream);
}

extern "C" int32_t vipant_gemm_bias_residual_bwd_e4m3(const uint16_t* dy, const uint16_t* w_t, const uint16_t* a, uint16_t* da,
                                                      float* dw, int64_t M, int64_t N, int64_t K, void* workspace,
                                                      size_t workspace_bytes, const vipant_fp8_plan* plan, void* stream) {
    VIPANT_REQUIRE(workspace_bytes >= vipant_gemm_tn_workspace_bytes(M, N, K), VIPANT_ENOWORKSPACE,
                   "gemm_bias_residual_bwd: workspace too small");
    // da[M, K] = dy[M, N] . W[N, K]   (w_t is W^T, [K, N]);  dW[N, K] = dy^T a
    TRY(nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, plan ? plan->dy_q : nullptr,
           plan ? plan->dy_scale : nullptr, dy, w_t, da, nullptr, nullptr, M, K, N, VIPANT_EPI_BF16, stream));
    if (tn8(plan, N) && K % 128 == 0 && plan->dy_q != nullptr) {
        const uint8_t *aq, *as, *bq, *bs;
        TRY(operand8(plan->dy_q, plan->dy_scale, false, dy, nullptr, nullptr, M, N, &aq, &as, stream));
        TRY(operand8(const_cast<uint8_t*>(plan->keep_q), const_cast<uint8_t*>(plan->keep_scale), true, a, plan->act_q, plan->act_scale, M,
                     K, &bq, &bs, stream));
        return vipant_gemm_tn_e4m3(aq, N, as, bq, K, bs, dw, K, M, N, K, 0, nullptr, workspace, workspace_bytes, stream);
    }
    return vipant_gemm_tn(dy, N, a, K, dw, K, M, N, K, 0, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int32_t vipant_gemm_bias_residual_bwd(const uint16_t* dy, const uint16_t* w_t, const uint16_t* a, uint16_t* da,
                                                 float* dw, int64_t M, int64_t N, int64_t K, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
    return vipant_gemm_bias_residual_bwd_e4m3(dy, w_t, a, da, dw, M, N, K, workspace, workspace_bytes, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------ K5: ln_2 + MLP
extern "C" int32_t vipant_ln_mlp_quickgelu_fwd_e4m3(const void* x, const uint16_t* add, void* x_out, const float* gamma,
                                                    const float* beta, const uint16_t* w_fc, const float* b_fc,
                                                    const uint16_t* w_proj, const float* b_proj, uint16_t* h, float* mean,
                                                    float* rstd, uint8_t* dcode, uint16_t* g, uint16_t* y, int64_t M, int64_t D,
                                                    const vipant_fp8_plan* plan, int32_t stream_flags, void* stream) {
    VIPANT_REQUIRE(M > 0 && D > 0 && D % 64 == 0, VIPANT_EBADSHAPE, "ln_mlp_quickgelu_fwd: bad shape M=%ld D=%ld", (long)M, (long)D);
    VIPANT_REQUIRE((add == nullptr) == (x_out == nullptr), VIPANT_EBADSHAPE, "ln_mlp_quickgelu_fwd: add and x_out go together");
    TRY(vipant_layernorm_fwd_e4m3(x, D, gamma, beta, h, nullptr, mean, rstd, M, D, add, x_out, plan ? plan->act_q : nullptr,
                                  plan ? plan->act_scale : nullptr, stream_flags, stream));
    // with a plan the c_fc epilogue leaves g's e4m3 form (block scales) in the plan's emit buffers and c_proj reads it from there: no
    // quantisation pass over the [M, 4D] activation; `g` / `dcode` may then be NULL (`running.recompute_mlp`: neither is kept)
    const bool emit = plan != nullptr && plan->emit_q != nullptr;
    // (g == NULL with dcode given, round 6: the e4m3 form and the derivative codes are all the backward will read)
    VIPANT_REQUIRE(emit || (g != nullptr && dcode != nullptr), VIPANT_EBADSHAPE, "ln_mlp_quickgelu_fwd: g and dcode are required");
    TRY(nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, plan ? plan->act_q : nullptr,
           plan ? plan->act_scale : nullptr, h, w_fc, g, b_fc, dcode, M, 4 * D, D, VIPANT_EPI_QUICKGELU_D8, stream,
           emit ? plan->emit_q : nullptr, emit ? plan->emit_scale : nullptr));
    return nt(plan, plan ? plan->w2_q : nullptr, plan ? plan->w2_scale : nullptr, emit ? plan->emit_q : nullptr,
              emit ? plan->emit_scale : nullptr, g, w_proj, y, b_proj, nullptr, M, D, 4 * D, VIPANT_EPI_BF16, stream);
}

extern "C" int32_t vipant_ln_mlp_quickgelu_fwd(const float* x, const uint16_t* add, float* x_out, const float* gamma,
                                               const float* beta, const uint16_t* w_fc, const float* b_fc,
                                               const uint16_t* w_proj, const float* b_proj, uint16_t* h, float* mean,
                                               float* rstd, uint8_t* dcode, uint16_t* g, uint16_t* y, int64_t M, int64_t D,
                                               void* stream) {
    return vipant_ln_mlp_quickgelu_fwd_e4m3(x, add, x_out, gamma, beta, w_fc, b_fc, w_proj, b_proj, h, mean, rstd, dcode, g, y, M, D,
                                            nullptr, 0, stream);
}

// The [M, 4D] activations (QuickGELU' codes, g) alone, from the saved LayerNorm output (`running.recompute_mlp`: they were not kept).
extern "C" int32_t vipant_mlp_quickgelu_recompute_e4m3(const uint16_t* h, const uint16_t* w_fc, const float* b_fc, uint8_t* dcode,
                                                       uint16_t* g, int64_t M, int64_t D, const vipant_fp8_plan* plan,
                                                       void* stream) {
    // h == NULL with a plan: the plan's act_q / act_scale already hold h's e4m3 form (kept from the forward's LayerNorm pass)
    // (emit_q given, round 6: g's e4m3 form -- what a forward that keeps it would have kept, byte for byte -- is left there; g may be NULL)
    const bool kept = plan != nullptr && h == nullptr;
    const bool emit = plan != nullptr && plan->emit_q != nullptr;
    return nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, kept ? plan->act_q : nullptr,
              kept ? plan->act_scale : nullptr, h, w_fc, g, b_fc, dcode, M, 4 * D, D, VIPANT_EPI_QUICKGELU_D8, stream,
              emit ? plan->emit_q : nullptr, emit ? plan->emit_scale : nullptr);
}

extern "C" int32_t vipant_mlp_quickgelu_recompute(const uint16_t* h, const uint16_t* w_fc, const float* b_fc, uint8_t* dcode,
                                                  uint16_t* g, int64_t M, int64_t D, void* stream) {
    return vipant_mlp_quickgelu_recompute_e4m3(h, w_fc, b_fc, dcode, g, M, D, nullptr, stream);
}

extern "C" int32_t vipant_ln_mlp_quickgelu_bwd_e4m3(const uint16_t* dy, const uint16_t* w_proj_t, const uint16_t* w_fc_t,
                                                    const uint8_t* dcode, const uint16_t* g, const uint16_t* h, const void* x,
                                                    const float* mean, const float* rstd, const float* gamma, float* dstream,
                                                    uint16_t* dx_bf16, uint16_t* du, uint16_t* dh, float* dw_proj, float* dw_fc,
                                                    float* db_fc, float* dgamma, float* dbeta, float* dx_colsum, int64_t M,
                                                    int64_t D, void* workspace, size_t workspace_bytes, const vipant_fp8_plan* plan,
                                                    int32_t stream_flags, void* stream) {
    VIPANT_REQUIRE(M > 0 && D > 0 && D % 64 == 0, VIPANT_EBADSHAPE, "ln_mlp_quickgelu_bwd: bad shape M=%ld D=%ld", (long)M, (long)D);
    VIPANT_REQUIRE(workspace_bytes >= vipant_block_workspace_bytes(M, D), VIPANT_ENOWORKSPACE,
                   "ln_mlp_quickgelu_bwd: workspace too small");
    // du = (dy . W_proj) * QuickGELU'(u), the derivative read from its 8-bit code;  dW_proj = dy^T g   (d b_proj is the column
    // sum the caller already has)
    // (VIPANT_STREAM_FEW_ROWS: the operator runs on a batch's read-out rows -- bf16 only: the e4m3 contractions have one kernel)
    const int32_t few = (plan == nullptr && (stream_flags & VIPANT_STREAM_FEW_ROWS)) ? VIPANT_EPI_FEW_ROWS : 0;
    const bool emit = plan != nullptr && plan->emit_q != nullptr;     // du's e4m3 form straight from the epilogue that makes du
    // (e4m3 weight gradients: nothing reads du's bf16 form any more -- c_fc^T and both users of du in the weight-gradient contraction
    // take the e4m3 form, the bias gradient rides on that contraction -- so it is not written: `du` may be NULL then)
    const bool du_q_only = tn8(plan, D) && plan->dy_q != nullptr;
    TRY(nt(plan, plan ? plan->w_q : nullptr, plan ? plan->w_scale : nullptr, plan ? plan->dy_q : nullptr,
           plan ? plan->dy_scale : nullptr, dy, w_proj_t, du_q_only ? nullptr : du, nullptr, const_cast<uint8_t*>(dcode), M, 4 * D, D,
           VIPANT_EPI_DQUICKGELU_D8 | few, stream, emit ? plan->emit_q : nullptr, emit ? plan->emit_scale : nullptr));
    // dh = du . W_fc;  dW_fc = du^T h, d b_fc = column sums of du
    // (Order, round 5: both input-gradient contractions first, then both weight gradients.  The NT kernels walk their tiles by
    // tickets and lose 1/256 of a launch per CU another stream holds; a weight-gradient launch is one wave of <= 256 long
    // workgroups and waits for a held CU.  The replica group's bucket all-reduce starts at a block boundary, i.e. right here: the
    // first 1.5 ms of a block's backward are now the two launches that tolerate it.  du is also read while it is still in the
    // 256 MB cache.)
    TRY(nt(plan, plan ? plan->w2_q : nullptr, plan ? plan->w2_scale : nullptr, emit ? plan->emit_q : nullptr,
           emit ? plan->emit_scale : nullptr, du, w_fc_t, dh, nullptr, nullptr, M, D, 4 * D, VIPANT_EPI_BF16 | few, stream));
    if (tn8(plan, D) && plan->dy_q != nullptr) {
        // (every operand's e4m3 form has served its NT contraction by now -- stream order -- and may change in place)
        const uint8_t *aq, *as, *bq, *bs;
        TRY(operand8(plan->dy_q, plan->dy_scale, false, dy, nullptr, nullptr, M, D, &aq, &as, stream));
        TRY(operand8(const_cast<uint8_t*>(plan->keep_q), const_cast<uint8_t*>(plan->keep_scale), true, g, plan->act_q, plan->act_scale, M,
                     4 * D, &bq, &bs, stream));
        TRY(vipant_gemm_tn_e4m3(aq, D, as, bq, 4 * D, bs, dw_proj, 4 * D, M, D, 4 * D, 0, nullptr, workspace, workspace_bytes, stream));
        TRY(operand8(plan->emit_q, plan->emit_scale, true /* the epilogue's form is block-uniform */, du, nullptr, nullptr, M, 4 * D, &aq,
                     &as, stream));
        TRY(operand8(const_cast<uint8_t*>(plan->keep2_q), const_cast<uint8_t*>(plan->keep2_scale), true, h, plan->act_q, plan->act_scale,
                     M, D, &bq, &bs, stream));
        TRY(vipant_gemm_tn_e4m3(aq, 4 * D, as, bq, D, bs, dw_fc, D, M, 4 * D, D, 0, db_fc, workspace, workspace_bytes, stream));
    } else {
        TRY(vipant_gemm_tn(dy, D, g, 4 * D, dw_proj, 4 * D, M, D, 4 * D, 0, nullptr, workspace, workspace_bytes, stream));
        TRY(vipant_gemm_tn(du, 4 * D, h, D, dw_fc, D, M, 4 * D, D, 0, db_fc, workspace, workspace_bytes, stream));
    }
    // ln_2 backward + residual-gradient add, in place; its dx is also d(out_proj output): dx_colsum = d out_proj.bias
    return ln_bwd(plan, dh, x, stream_flags, mean, rstd, gamma, dstream, dx_bf16, dgamma, dbeta, dx_colsum, M, D, workspace,
                  workspace_bytes, stream);
}

extern "C" int32_t vipant_ln_mlp_quickgelu_bwd(const uint16_t* dy, const uint16_t* w_proj_t, const uint16_t* w_fc_t,
                                               const uint8_t* dcode, const uint16_t* g, const uint16_t* h, const float* x,
                                               const float* mean, const float* rstd, const float* gamma, float* dstream,
                                               uint16_t* dx_bf16, uint16_t* du, uint16_t* dh, float* dw_proj, float* dw_fc,
                                               float* db_fc, float* dgamma, float* dbeta, float* dx_colsum, int64_t M,
                                               int64_t D, void* workspace, size_t workspace_bytes, void* stream) {
    return vipant_ln_mlp_quickgelu_bwd_e4m3(dy, w_proj_t, w_fc_t, dcode, g, h, x, mean, rstd, gamma, dstream, dx_bf16, du, dh, dw_proj,
                                            dw_fc, db_fc, dgamma, dbeta, dx_colsum, M, D, workspace, workspace_bytes, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------ K1: patch embedding + ln_pre
extern "C" int32_t vipant_patch_embed_ln_fwd(const float* x, const float* conv_w, const float* cls, const float* pos,
                                             const float* gamma, const float* beta, uint16_t* w_eff, uint16_t* patches,
                                             float* pe, float* tokens, float* out, float* mean, float* rstd, int64_t b,
                                             int64_t C, int64_t T, int64_t F, int64_t Dw, int64_t Cw, int64_t ph, int64_t pw,
                                             int64_t sh, int64_t sw, int32_t mean_channels, void* stream) {
    const int64_t nrow = (T - ph) / sh + 1, ncol = (F - pw) / sw + 1;
    const int64_t P = nrow * ncol, S = P + 1;
    const int64_t kcols = (mean_channels ? 1 : Cw) * ph * pw;
    VIPANT_REQUIRE(nrow > 0 && ncol > 0 && kcols % 64 == 0, VIPANT_EBADSHAPE, "patch_embed_ln_fwd: bad geometry");
    TRY(vipant_conv_weight_prep(conv_w, w_eff, Dw, Cw, ph * pw, mean_channels, stream));
    TRY(vipant_im2col(x, patches, b, C, T, F, ph, pw, sh, sw, stream));
    // the contraction writes the patch rows of the token matrix itself (+ pos); `pe` is no longer used (kept in the signature)
    (void)pe;
    TRY(vipant_gemm_nt_tokens(patches, kcols, w_eff, kcols, tokens, pos, b, P, Dw, kcols, stream));
    TRY(vipant_tokens_cls_rows(cls, pos, tokens, b, S, Dw, stream));
    return vipant_layernorm_fwd(tokens, Dw, gamma, beta, nullptr, out, mean, rstd, b * S, Dw, nullptr, nullptr, stream);
}

extern "C" size_t vipant_patch_embed_ln_bwd_workspace_bytes(int64_t b, int64_t P, int64_t Dw, int64_t kcols) {
    return max_sz(vipant_gemm_tn_workspace_bytes(b * P, Dw, kcols), vipant_layernorm_bwd_workspace_bytes(b * (P + 1), Dw));
}

extern "C" int32_t vipant_patch_embed_ln_bwd(const void* dout, int32_t dout_bf16, const float* tokens, const float* mean, const float* rstd,
                                             const float* gamma, const uint16_t* patches, float* dtokens, uint16_t* dpatches,
                                             float* dw_eff, float* dconv, float* dcls, float* dpos, float* dgamma,
                                             float* dbeta, int64_t b, int64_t P, int64_t Dw, int64_t Cw, int64_t khw,
                                             int32_t mean_channels, void* workspace, size_t workspace_bytes, void* stream) {
    const int64_t kcols = (mean_channels ? 1 : Cw) * khw;
    VIPANT_REQUIRE(workspace_bytes >= vipant_patch_embed_ln_bwd_workspace_bytes(b, P, Dw, kcols), VIPANT_ENOWORKSPACE,
                   "patch_embed_ln_bwd: workspace too small");
    TRY(vipant_layernorm_bwd(dout, dout_bf16 ? 0 : VIPANT_LN_DY_F32, tokens, Dw, mean, rstd, gamma, nullptr, dtokens, Dw, nullptr, dgamma,
                             dbeta, nullptr, 0, b * (P + 1), Dw, workspace, workspace_bytes, stream));
    TRY(vipant_assemble_tokens_bwd(dtokens, dpatches, dcls, dpos, 0, b, P, Dw, stream));    // dpos must arrive zeroed
    TRY(vipant_gemm_tn(dpatches, Dw, patches, kcols, dw_eff, kcols, b * P, Dw, kcols, 0, nullptr, workspace, workspace_bytes,
                       stream));
    if (mean_channels) return vipant_conv_weight_grad(dw_eff, dconv, Dw, Cw, khw, 0, stream);
    return VIPANT_OK;       // dconv == dw_eff viewed as [Dw, Cw, kh, kw]
}

// ------------------------------------------------------------------------------------------------ K6 / K7: read-outs
// LN(x[b, row_b]) @ proj (+ L2 normalisation): row_b = 0 (cls, ViTPostEncoder) when idx == NULL, else idx[b] (EOT, GPTPostEncoder).
extern "C" int32_t vipant_cls_ln_proj_l2norm_fwd(const float* x, const int64_t* idx, const float* gamma, const float* beta,
                                                 const uint16_t* proj_t, float* rows, uint16_t* y, float* mean, float* rstd,
                                                 float* feat, float* out, float* norm, int64_t batch, int64_t S, int64_t D,
                                                 int64_t E, int32_t normalized, void* stream) {
    const float* src = x;
    int64_t ld = S * D;                       // cls row of every sample: strided view, no copy
    if (idx != nullptr) {
        VIPANT_REQUIRE(rows != nullptr, VIPANT_EBADSHAPE, "cls_ln_proj_l2norm_fwd: gathered read-out needs `rows`");
        TRY(vipant_gather_rows(x, idx, rows, batch, S, D, stream));
        src = rows; ld = D;
    }
    TRY(vipant_layernorm_fwd(src, ld, gamma, beta, y, nullptr, mean, rstd, batch, D, nullptr, nullptr, stream));
    TRY(vipant_gemm_nt(y, D, proj_t, D, feat, E, nullptr, nullptr, 1.0f, batch, E, D, VIPANT_EPI_F32 | VIPANT_EPI_FEW_ROWS, stream));
    if (normalized) return vipant_l2norm_fwd(feat, out, norm, batch, E, stream);
    return VIPANT_OK;
}

extern "C" int32_t vipant_eot_ln_proj_l2norm_fwd(const float* x, const int64_t* eot, const float* gamma, const float* beta,
                                                 const uint16_t* proj_t, float* rows, uint16_t* y, float* mean, float* rstd,
                                                 float* feat, float* out, float* norm, int64_t batch, int64_t L, int64_t D,
                                                 int64_t E, int32_t normalized, void* stream) {
    VIPANT_REQUIRE(eot != nullptr, VIPANT_EBADSHAPE, "eot_ln_proj_l2norm_fwd: eot indices are required");
    return vipant_cls_ln_proj_l2norm_fwd(x, eot, gamma, beta, proj_t, rows, y, mean, rstd, feat, out, norm, batch, L, D, E,
                                         normalized, stream);
}

extern "C" size_t vipant_cls_ln_proj_l2norm_bwd_workspace_bytes(int64_t batch, int64_t D, int64_t E) {
    return max_sz(vipant_gemm_tn_workspace_bytes(batch, D, E), vipant_layernorm_bwd_workspace_bytes(batch, D));
}

// dx (token-major, zeroed by the caller) receives the read-out rows' gradient; dproj [D, E] = y^T dfeat.
extern "C" int32_t vipant_cls_ln_proj_l2norm_bwd(const float* dout, const float* out, const float* norm, const float* x,
                                                 const int64_t* idx, const float* rows, const uint16_t* y, const float* mean,
                                                 const float* rstd, const float* gamma, const uint16_t* proj, uint16_t* dfeat,
                                                 uint16_t* dy, float* drows, float* dx, float* dproj, float* dgamma,
                                                 float* dbeta, int64_t batch, int64_t S, int64_t D, int64_t E,
                                                 int32_t normalized, void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(workspace_bytes >= vipant_cls_ln_proj_l2norm_bwd_workspace_bytes(batch, D, E), VIPANT_ENOWORKSPACE,
                   "cls_ln_proj_l2norm_bwd: workspace too small");
    if (normalized) TRY(vipant_l2norm_bwd(dout, out, norm, nullptr, dfeat, batch, E, stream));
    else TRY(vipant_cast_bf16(dout, dfeat, nullptr, 1, batch * E, stream));
    TRY(vipant_gemm_nt(dfeat, E, proj, E, dy, D, nullptr, nullptr, 1.0f, batch, D, E, VIPANT_EPI_BF16 | VIPANT_EPI_FEW_ROWS,
                       stream));   // dy = dfeat . proj^T
    TRY(vipant_gemm_tn(y, D, dfeat, E, dproj, E, batch, D, E, 0, nullptr, workspace, workspace_bytes, stream));
    if (idx == nullptr) {
        // cls rows: in place in the token-major dx (row stride S * D), or -- `drows` given -- as compact rows [batch, D]: the caller
        // then scatters them itself (the stack's backward wants them as bf16 rows of its stream gradient, not as a dense fp32 matrix)
        if (drows != nullptr)
            return vipant_layernorm_bwd(dy, 0, x, S * D, mean, rstd, gamma, nullptr, drows, D, nullptr, dgamma, dbeta, nullptr, 0,
                                        batch, D, workspace, workspace_bytes, stream);
        return vipant_layernorm_bwd(dy, 0, x, S * D, mean, rstd, gamma, nullptr, dx, S * D, nullptr, dgamma, dbeta, nullptr, 0,
                                    batch, D, workspace, workspace_bytes, stream);
    }
    TRY(vipant_layernorm_bwd(dy, 0, rows, D, mean, rstd, gamma, nullptr, drows, D, nullptr, dgamma, dbeta, nullptr, 0, batch, D,
                             workspace, workspace_bytes, stream));
    if (dx == nullptr) return VIPANT_OK;                  // compact form: the caller scatters drows
    return vipant_scatter_rows(drows, idx, dx, batch, S, D, stream);
}

// GPTPreEncoder.forward (val.py:109-122): the export-set name of vipant_embed_tokens.
extern "C" int32_t vipant_embed_gather_pos_fwd(const int64_t* tokens, const float* table, const float* pos, float* x,
                                               int64_t* eot, int64_t b, int64_t L, int64_t D, void* stream) {
    return vipant_embed_tokens(tokens, table, pos, x, eot, b, L, D, stream);
}
