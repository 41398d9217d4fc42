// The last block on its read-out rows, attention half in ALGEBRAIC form (round 4; DESIGN.md section 5, "the last block").
//
// ViTPostEncoder / GPTPostEncoder read ONE row per item (cvap/module/val.py:288-289, 143-145), so the last block's attention has one
// query per (item, head).  Round 3 still projected every token to K and V (a [M, 2D] contraction, its dX and its dW: 1.0 ms per step)
// for those queries to look at.  With one query the projections fold into the query side:
//     scores_j   = q_h . (W_k,h h1_j + b_k,h) / 8 = (W_k,h^T q_h) . h1_j / 8 + const          (the constant cancels in the softmax)
//     o_h        = sum_j p_j (W_v,h h1_j + b_v,h)  = W_v,h (sum_j p_j h1_j) + b_v,h            (sum_j p_j = 1)
// i.e. per (item, head) a query-like vector qk_h = W_k,h^T q_h of width D and a context ctx_h = sum_j p_j h1_j of width D, both
// against the LayerNorm output h1 itself.  The projections that remain are [batch * H, D] x [D, D] contractions (block-sparse head
// expansion of the `batch` rows: `vipant_head_expand` / `vipant_head_extract`).  Backward: with dctx_h = W_v,h^T do_h,
//     dp_j = dctx_h . h1_j,  delta = dctx_h . ctx_h,  ds_j = p_j (dp_j - delta) / 8,
//     dh1_j = sum_h p_j,h dctx_h + ds_j,h qk_h        (rank-2H update per token),     dqk_h = sum_j ds_j,h h1_j.
// Both kernels stream h1 (fwd: read; bwd: read + write dh1) in place of the K / V projection, its dX, its dW and the two one-query
// kernels.  One workgroup of four waves per item.  The dot products over D (scores, dp) and the rank-2H update that is dh1 run on
// v_mfma_f32_16x16x32_bf16 with the heads padded to 16; the two sums over tokens (contexts, dqk) are fp32 multiply-adds on the VALU,
// a wave owning H / 4 heads.
#include "common.h"

namespace {

constexpr float LOG2E_ = 1.4426950408889634f;

// sum over the 64 lanes, returned in every lane (wave-uniform: through an SGPR)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad xor 1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad xor 2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    // every lane of a 16-lane row now holds the row's sum: rows 1, 3 += rows 0, 2; rows 2, 3 += row 1; lane 63 has all four
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// a lane's EPL elements of a D-wide bf16 row: 4 consecutive elements at 256 t + 4 lane, t = 0 .. EPL / 4 - 1 (512-byte wave loads)
template <int EPL>
struct RawRow { bf16x4 v[EPL / 4]; };

template <int EPL>
__device__ __forceinline__ RawRow<EPL> load_raw(const bf16_t* row, int lane) {
    RawRow<EPL> r;
#pragma unroll
    for (int t = 0; t < EPL / 4; ++t) r.v[t] = *(const bf16x4*)(row + 256 * t + 4 * lane);
    return r;
}
template <int EPL>
__device__ __forceinline__ void widen(const RawRow<EPL>& r, float (&x)[EPL]) {
#pragma unroll
    for (int t = 0; t < EPL / 4; ++t) {
        x[4 * t] = (float)r.v[t][0]; x[4 * t + 1] = (float)r.v[t][1]; x[4 * t + 2] = (float)r.v[t][2]; x[4 * t + 3] = (float)r.v[t][3];
    }
}
template <int EPL>
__device__ __forceinline__ void load_row(const bf16_t* row, int lane, float (&x)[EPL]) { widen<EPL>(load_raw<EPL>(row, lane), x); }
template <int EPL>
__device__ __forceinline__ void store_row(bf16_t* row, int lane, const float (&x)[EPL]) {
#pragma unroll
    for (int t = 0; t < EPL / 4; ++t)
        *(bf16x4*)(row + 256 * t + 4 * lane) = f32x4_to_bf16x4(f32x4{x[4 * t], x[4 * t + 1], x[4 * t + 2], x[4 * t + 3]});
}

__device__ __forceinline__ int key_limit(const int64_t* idx, int item, int S, int causal) {
    if (!causal || idx == nullptr) return S;
    const int64_t v = idx[item];
    return (int)(v < 0 ? 0 : (v >= S ? S - 1 : v)) + 1;
}

__device__ __forceinline__ float wave_max_dpp(float v) {
#define VIPANT_MAX_DPP(ctrl, rmask)                                                                                                       \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), ctrl,     \
                                                                         rmask, 0xF, false)))
    VIPANT_MAX_DPP(0xB1, 0xF);
    VIPANT_MAX_DPP(0x4E, 0xF);
    VIPANT_MAX_DPP(0x141, 0xF);
    VIPANT_MAX_DPP(0x140, 0xF);
    VIPANT_MAX_DPP(0x142, 0xA);
    VIPANT_MAX_DPP(0x143, 0xC);
#undef VIPANT_MAX_DPP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The dot products of 16 tokens with all heads on the matrix pipe: acc[head = lane & 15][token 4 (lane >> 4) + i] =
// sum_k h1[token][k] b[head][k].  A fragments straight from the row-major h1 rows (16 bytes per lane and k-step), B fragments `bq`
// (the heads' vectors, resident).  `arow`: this lane's token row + 8 (lane >> 4) elements.
template <int KS>
__device__ __forceinline__ f32x4 dots_tile(const bf16_t* arow, const bf16x8 (&bq)[KS]) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int HALF = KS / 2;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        bf16x8 af[HALF];
#pragma unroll
        for (int s = 0; s < HALF; ++s) af[s] = *(const bf16x8*)(arow + 32 * (part * HALF + s));
#pragma unroll
        for (int s = 0; s < HALF; ++s) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], bq[part * HALF + s], a, 0, 0, 0);
    }
    return a;
}
// the same with the B fragments in LDS (`bqs`: [KS][64 lanes] fragments, this lane's at bqs[64 s])
template <int KS>
__device__ __forceinline__ f32x4 dots_tile_lds(const bf16_t* arow, const bf16x8* bqs) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int HALF = KS / 2;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        bf16x8 af[HALF];
#pragma unroll
        for (int s = 0; s < HALF; ++s) af[s] = *(const bf16x8*)(arow + 32 * (part * HALF + s));
#pragma unroll
        for (int s = 0; s < HALF; ++s) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s], bqs[64 * (part * HALF + s)], a, 0, 0, 0);
    }
    return a;
}

// ctx[item, h, :] = sum_j softmax_j(qk[item, h, :] . h1[item, j, :] / 8) h1[item, j, :];  probs[item, h, j] = that softmax (fp32).
// One workgroup per item, rounds of 64 tokens: (A) wave w takes the scores of tokens 16 w .. 16 w + 15 of the round against ALL heads
// on the matrix pipe and leaves them in LDS; (B) wave w owns heads HPW w .. HPW w + HPW - 1: one lane per token of the round takes the
// round's maximum, rescales if it moved, exponentiates; (C) the weighted sum of the 64 rows for the wave's heads on the VALU
// (2 D multiply-adds per head and token, the weights broadcast from LDS).  Raw scores wait in LDS until maximum and sum are final.
template <int EPL, int HPW, int MINB>
__global__ __launch_bounds__(256, MINB) void rows_ctx_fwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ h1,
                                                                 const int64_t* __restrict__ idx, bf16_t* __restrict__ ctx,
                                                                 float* __restrict__ probs, int S, int causal) {
    constexpr int D = EPL * 64, NH = 4 * HPW, KS = D / 32, PF = 8;
    extern __shared__ float sm[];     // [NH][Sp] raw scores (exp2 domain) | [4 waves][HPW][64] weights of the round | B fragments
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const int item = blockIdx.x;
    const int nkeys = key_limit(idx, item, S, causal);
    const int Sp = (S + 63) & ~63;
    float* sc = sm;
    float* pp = sm + NH * Sp + wave * HPW * 64;
    bf16x8* bqs = (bf16x8*)(sm + NH * Sp + 4 * HPW * 64) + lane;     // [KS][64]: the heads' vectors as B fragments, lane-linear
    const int64_t head0 = (int64_t)item * NH + HPW * wave;
    const bf16_t* rows = h1 + (int64_t)item * S * D;
    {
        const bf16_t* qrow = qk + ((int64_t)item * NH + (r < NH ? r : 0)) * D + 8 * g;
        for (int s = wave; s < KS; s += 4) bqs[64 * s] = r < NH ? *(const bf16x8*)(qrow + 32 * s) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __syncthreads();
    float acc[HPW][EPL], m[HPW], lsum[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[h][e] = 0.f;
        m[h] = -INFINITY; lsum[h] = 0.f;
    }
    RawRow<EPL> nxt[PF];                                // the rows of the next PF tokens, in flight while the current PF are used
#pragma unroll
    for (int t = 0; t < PF; ++t) nxt[t] = load_raw<EPL>(rows + (int64_t)(t < S ? t : S - 1) * D, lane);
    for (int j0 = 0; j0 < nkeys; j0 += 64) {
        {   // (A)
            const int tok = j0 + 16 * wave + r;
            const f32x4 a = dots_tile_lds<KS>(rows + (int64_t)(tok < S ? tok : S - 1) * D + 8 * g, bqs);
            if (r < NH) *(f32x4*)(sc + r * Sp + j0 + 16 * wave + 4 * g) = a * (0.125f * LOG2E_);
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < HPW; ++h) {   // (B)
            const int j = j0 + lane;
            const float sv = j < nkeys ? sc[(HPW * wave + h) * Sp + j] : -INFINITY;
            const float rmax = wave_max_dpp(sv);
            if (rmax > m[h]) {                          // wave-uniform
                const float f = __builtin_amdgcn_exp2f(m[h] - rmax);
                lsum[h] *= f;
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[h][e] *= f;
                m[h] = rmax;
            }
            const float pj = __builtin_amdgcn_exp2f(sv - m[h]);
            lsum[h] += pj;                              // per-lane share of the sum; added up across lanes at the end
            pp[h * 64 + lane] = pj;
        }
        for (int t0 = 0; t0 < 64 && j0 + t0 < nkeys; t0 += PF) {   // (C)
            f32x4 pw[HPW][PF / 4];
#pragma unroll
            for (int h = 0; h < HPW; ++h)
#pragma unroll
                for (int u = 0; u < PF / 4; ++u) pw[h][u] = *(const f32x4*)(pp + h * 64 + t0 + 4 * u);
#pragma unroll
            for (int t = 0; t < PF; ++t) {
                float x[EPL];
                widen<EPL>(nxt[t], x);
                const int jn = j0 + t0 + t + PF;
                nxt[t] = load_raw<EPL>(rows + (int64_t)(jn < S ? jn : S - 1) * D, lane);
#pragma unroll
                for (int h = 0; h < HPW; ++h)
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc[h][e] = __builtin_fmaf(pw[h][t >> 2][t & 3], x[e], acc[h][e]);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const float inv = 1.0f / wave_sum_dpp(lsum[h]);
        float o[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) o[e] = acc[h][e] * inv;
        store_row<EPL>(ctx + (head0 + h) * D, lane, o);
        float* pout = probs + (head0 + h) * S;
        for (int j = lane; j < S; j += 64)
            pout[j] = j < nkeys ? __builtin_amdgcn_exp2f(sc[(HPW * wave + h) * Sp + j] - m[h]) * inv : 0.f;
    }
}

// dh1[item, j, :] = sum_h p_j,h dctx_h + ds_j,h qk_h;  dqk[item, h, :] = sum_j ds_j,h h1_j;  ds_j,h = p_j,h (dctx_h . h1_j - dctx_h . ctx_h) / 8.
// Rounds of 64 tokens as in the forward: (A) dctx_h . h1_j for 16 tokens x all heads per wave on the matrix pipe; (B) one lane per
// token: ds for the wave's heads, left in LDS as fp32 (for C) and, with p, as the bf16 row [p_0 .. p_15 | ds_0 .. ds_15] of the
// token (for D); (C) dqk of the wave's heads on the VALU; (D) dh1 of the wave's 16 tokens as ONE matrix product per 16 columns:
// (dh1 tile)^T [16 columns x 16 tokens] = [dctx | qk]^T [16 columns x 32] . [p | ds]^T [32 x 16 tokens], the left operand resident in
// LDS for the item ([D][32] bf16, 16-byte chunks XOR-ed with (column >> 2) & 3: conflict-free 128-bit reads).
template <int EPL, int HPW, int MINB>
__global__ __launch_bounds__(256, MINB) void rows_ctx_bwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ dctx,
                                                                 const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ h1,
                                                                 const int64_t* __restrict__ idx, const float* __restrict__ probs,
                                                                 bf16_t* __restrict__ dh1, bf16_t* __restrict__ dqk, int S, int causal) {
    constexpr int D = EPL * 64, NH = 4 * HPW, KS = D / 32, PF = EPL <= 8 ? 8 : 4;
    extern __shared__ float sm[];
    bf16_t* dcq = (bf16_t*)sm;                          // [D][32]: 64 D bytes
    bf16_t* pds = dcq + D * 32;                         // [64 tokens][32]: 4 KiB
    float* dpb = (float*)(pds + 64 * 32);               // [16 heads][64 tokens] fp32: 4 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    float* dsb = dpb + 16 * 64 + wave * HPW * 64;       // [4 waves][HPW][64] fp32
    const int item = blockIdx.x;
    const int nkeys = key_limit(idx, item, S, causal);
    const int64_t head0 = (int64_t)item * NH + HPW * wave;
    const bf16_t* rows = h1 + (int64_t)item * S * D;
    bf16_t* drows = dh1 + (int64_t)item * S * D;
    // the item's [dctx | qk]^T and the zero padding of the token rows
    for (int i = threadIdx.x; i < (D * 32 + 64 * 32) / 8; i += 256) ((u32x4*)dcq)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int i = threadIdx.x; i < NH * D; i += 256) {
        const int hh = i / D, n = i - hh * D;
        const int sw = (n >> 2) & 3;
        dcq[n * 32 + 8 * ((hh >> 3) ^ sw) + (hh & 7)] = dctx[((int64_t)item * NH + hh) * D + n];
        dcq[n * 32 + 8 * ((2 + (hh >> 3)) ^ sw) + (hh & 7)] = qk[((int64_t)item * NH + hh) * D + n];
    }
    bf16x8 bq[KS];
    {
        const bf16_t* qrow = dctx + ((int64_t)item * NH + (r < NH ? r : 0)) * D + 8 * g;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bq[s] = *(const bf16x8*)(qrow + 32 * s);
            if (r >= NH) bq[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    float acc[HPW][EPL], delta[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        float dc[EPL], c[EPL];
        load_row<EPL>(dctx + (head0 + h) * D, lane, dc);
        load_row<EPL>(ctx + (head0 + h) * D, lane, c);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) { d = __builtin_fmaf(dc[e], c[e], d); acc[h][e] = 0.f; }
        delta[h] = wave_sum_dpp(d);
    }
    RawRow<EPL> nxt[PF];                                // the rows of the next PF tokens, in flight while the current PF are used
#pragma unroll
    for (int t = 0; t < PF; ++t) nxt[t] = load_raw<EPL>(rows + (int64_t)(t < S ? t : S - 1) * D, lane);
    __syncthreads();
    for (int j0 = 0; j0 < S; j0 += 64) {
        const bool live = j0 < nkeys;                   // rounds behind a causal limit: zeros for dh1, nothing else
        if (live) {   // (A)
            const int tok = j0 + 16 * wave + r;
            const f32x4 a = dots_tile<KS>(rows + (int64_t)(tok < S ? tok : S - 1) * D + 8 * g, bq);
            *(f32x4*)(dpb + r * 64 + 16 * wave + 4 * g) = a;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < HPW; ++h) {   // (B)
            const int j = j0 + lane, hg = HPW * wave + h;
            const float pj = j < nkeys ? probs[(head0 + h) * S + j] : 0.f;
            const float ds = live ? pj * (dpb[hg * 64 + lane] - delta[h]) * 0.125f : 0.f;
            dsb[h * 64 + lane] = ds;
            const int sw = (lane >> 2) & 3;
            pds[lane * 32 + 8 * ((hg >> 3) ^ sw) + (hg & 7)] = (bf16_t)pj;
            pds[lane * 32 + 8 * ((2 + (hg >> 3)) ^ sw) + (hg & 7)] = (bf16_t)ds;
        }
        __syncthreads();
        if (live) {
            for (int t0 = 0; t0 < 64 && j0 + t0 < nkeys; t0 += PF) {   // (C)
                f32x4 dw[HPW][PF / 4];
#pragma unroll
                for (int h = 0; h < HPW; ++h)
#pragma unroll
                    for (int u = 0; u < PF / 4; ++u) dw[h][u] = *(const f32x4*)(dsb + h * 64 + t0 + 4 * u);
#pragma unroll
                for (int t = 0; t < PF; ++t) {
                    float x[EPL];
                    widen<EPL>(nxt[t], x);
                    const int jn = j0 + t0 + t + PF;
                    nxt[t] = load_raw<EPL>(rows + (int64_t)(jn < S ? jn : S - 1) * D, lane);
#pragma unroll
                    for (int h = 0; h < HPW; ++h)
#pragma unroll
                        for (int e = 0; e < EPL; ++e) acc[h][e] = __builtin_fmaf(dw[h][t >> 2][t & 3], x[e], acc[h][e]);
                }
            }
        }
        {   // (D)
            const int tl = 16 * wave + r, tok = j0 + tl;
            const bf16x8 bt = *(const bf16x8*)(pds + tl * 32 + 8 * (g ^ ((tl >> 2) & 3)));
            bf16_t* orow = drows + (int64_t)tok * D + 4 * g;
#pragma unroll 8
            for (int mt = 0; mt < D / 16; ++mt) {
                const int n = 16 * mt + r;
                const bf16x8 at = *(const bf16x8*)(dcq + n * 32 + 8 * (g ^ ((n >> 2) & 3)));
                const f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at, bt, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (tok < S) *(bf16x4*)(orow + 16 * mt) = f32x4_to_bf16x4(c);
            }
        }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h) store_row<EPL>(dqk + (head0 + h) * D, lane, acc[h]);
}

// out[(i, h), :] = 0 except columns 64 h .. 64 h + 63 = rows[i, 64 h ..]: the `batch` rows as block-sparse [batch * H, D] operand
__global__ __launch_bounds__(256) void head_expand_kernel(const bf16_t* __restrict__ rows, bf16_t* __restrict__ out, int64_t n, int H) {
    const int D = H * 64, cpr = D / 8;                  // 16-byte chunks per row
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n * H * cpr; c += (int64_t)gridDim.x * 256) {
        const int64_t r = c / cpr;
        const int col = (int)(c - r * cpr) * 8, h = (int)(r % H);
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (col / 64 == h) v = *(const u32x4*)(rows + (r / H) * D + col);
        *(u32x4*)(out + r * D + col) = v;
    }
}

// rows[i, 64 h + c] = full[(i, h), 64 h + c] (+ bias[64 h + c]): the diagonal blocks of a [batch * H, D] product, as bf16
template <typename T>
__global__ __launch_bounds__(256) void head_extract_kernel(const T* __restrict__ full, const float* __restrict__ bias,
                                                           bf16_t* __restrict__ rows, int64_t n, int H) {
    const int D = H * 64;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n * D / 4; c += (int64_t)gridDim.x * 256) {
        const int64_t i = (c * 4) / D;
        const int col = (int)(c * 4 - i * D), h = col / 64;
        const T* src = full + (i * H + h) * D + col;
        f32x4 v = f32x4{(float)src[0], (float)src[1], (float)src[2], (float)src[3]};
        if (bias != nullptr) v += *(const f32x4*)(bias + col);
        *(bf16x4*)(rows + i * D + col) = f32x4_to_bf16x4(v);
    }
}

int32_t check_ctx(int64_t batch, int64_t S, int64_t H) {
    VIPANT_REQUIRE(batch > 0 && S > 0 && S <= 2048 && (H == 8 || H == 12 || H == 16), VIPANT_EBADSHAPE,
                   "rows_ctx: batch=%ld S=%ld H=%ld (heads of 64: H = 8, 12 or 16, i.e. width 512, 768 or 1024; S <= 2048)", (long)batch,
                   (long)S, (long)H);
    return VIPANT_OK;
}

template <int EPL, int HPW, int MINB>
int32_t launch_fwd(const bf16_t* qk, const bf16_t* h1, const int64_t* idx, bf16_t* ctx, float* probs, int batch, int S, int causal,
                   hipStream_t st) {
    const int lds = (4 * HPW * ((S + 63) & ~63) + 4 * HPW * 64) * (int)sizeof(float) + EPL * 2 * 64 * 16;
    static int configured = 0;
    if (lds > configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)rows_ctx_fwd_kernel<EPL, HPW, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = lds;
    }
    hipLaunchKernelGGL((rows_ctx_fwd_kernel<EPL, HPW, MINB>), dim3((unsigned)batch), dim3(256), lds, st, qk, h1, idx, ctx, probs, S, causal);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int EPL, int HPW, int MINB>
int32_t launch_bwd(const bf16_t* qk, const bf16_t* dctx, const bf16_t* ctx, const bf16_t* h1, const int64_t* idx, const float* probs,
                   bf16_t* dh1, bf16_t* dqk, int batch, int S, int causal, hipStream_t st) {
    constexpr int lds = EPL * 64 * 64 + 64 * 64 + 16 * 64 * 4 + 4 * HPW * 64 * 4;
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)rows_ctx_bwd_kernel<EPL, HPW, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    hipLaunchKernelGGL((rows_ctx_bwd_kernel<EPL, HPW, MINB>), dim3((unsigned)batch), dim3(256), lds, st, qk, dctx, ctx, h1, idx, probs, dh1,
                       dqk, S, causal);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

extern "C" int32_t vipant_rows_ctx_fwd(const uint16_t* qk, const uint16_t* h1, const int64_t* idx, uint16_t* ctx, float* probs,
                                       int64_t batch, int64_t S, int64_t H, int32_t causal, void* stream) {
    if (int32_t e = check_ctx(batch, S, H)) return e;
    VIPANT_REQUIRE(qk != nullptr && h1 != nullptr && ctx != nullptr && probs != nullptr, VIPANT_EBADSHAPE, "rows_ctx_fwd: null operand");
    VIPANT_REQUIRE((uintptr_t)qk % 16 == 0 && (uintptr_t)h1 % 16 == 0 && (uintptr_t)ctx % 16 == 0, VIPANT_EALIGN,
                   "rows_ctx_fwd: operands must be 16-byte aligned");
    const bf16_t *a = (const bf16_t*)qk, *b = (const bf16_t*)h1;
    if (H == 12) return launch_fwd<12, 3, 2>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, (hipStream_t)stream);
    if (H == 16) return launch_fwd<16, 4, 1>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, (hipStream_t)stream);
    return launch_fwd<8, 2, 2>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, (hipStream_t)stream);
}

extern "C" int32_t vipant_rows_ctx_bwd(const uint16_t* qk, const uint16_t* dctx, const uint16_t* ctx, const uint16_t* h1,
                                       const int64_t* idx, const float* probs, uint16_t* dh1, uint16_t* dqk, int64_t batch, int64_t S,
                                       int64_t H, int32_t causal, void* stream) {
    if (int32_t e = check_ctx(batch, S, H)) return e;
    VIPANT_REQUIRE(qk != nullptr && dctx != nullptr && ctx != nullptr && h1 != nullptr && probs != nullptr && dh1 != nullptr && dqk != nullptr,
                   VIPANT_EBADSHAPE, "rows_ctx_bwd: null operand");
    VIPANT_REQUIRE((uintptr_t)qk % 16 == 0 && (uintptr_t)dctx % 16 == 0 && (uintptr_t)ctx % 16 == 0 && (uintptr_t)h1 % 16 == 0 &&
                   (uintptr_t)dh1 % 16 == 0 && (uintptr_t)dqk % 16 == 0, VIPANT_EALIGN, "rows_ctx_bwd: operands must be 16-byte aligned");
    const bf16_t *a = (const bf16_t*)qk, *b = (const bf16_t*)dctx, *c = (const bf16_t*)ctx, *d = (const bf16_t*)h1;
    hipStream_t st = (hipStream_t)stream;
    if (H == 12) return launch_bwd<12, 3, 2>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, st);
    if (H == 16) return launch_bwd<16, 4, 1>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, st);
    return launch_bwd<8, 2, 2>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, st);
}

extern "C" int32_t vipant_head_expand(const uint16_t* rows, uint16_t* out, int64_t n, int64_t H, void* stream) {
    VIPANT_REQUIRE(rows != nullptr && out != nullptr && n > 0 && H > 0, VIPANT_EBADSHAPE, "head_expand: bad arguments");
    VIPANT_REQUIRE((uintptr_t)rows % 16 == 0 && (uintptr_t)out % 16 == 0, VIPANT_EALIGN, "head_expand: operands must be 16-byte aligned");
    const int64_t chunks = n * H * (H * 8);
    const int64_t g = ceil_div(chunks, 256);
    hipLaunchKernelGGL(head_expand_kernel, dim3((unsigned)(g < 8192 ? g : 8192)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)rows,
                       (bf16_t*)out, n, (int)H);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_head_extract(const void* full, int32_t full_is_f32, const float* bias, uint16_t* rows, int64_t n, int64_t H,
                                       void* stream) {
    VIPANT_REQUIRE(full != nullptr && rows != nullptr && n > 0 && H > 0, VIPANT_EBADSHAPE, "head_extract: bad arguments");
    const int64_t g = ceil_div(n * H * 64 / 4, 256);
    const unsigned grid = (unsigned)(g < 4096 ? g : 4096);
    if (full_is_f32)
        hipLaunchKernelGGL(head_extract_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)full, bias, (bf16_t*)rows, n, (int)H);
    else
        hipLaunchKernelGGL(head_extract_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)full, bias, (bf16_t*)rows, n, (int)H);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}
