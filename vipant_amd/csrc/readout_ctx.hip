// The last block on its read-out rows, attention half in ALGEBRAIC form (round 4; DESIGN.md section 5, "the last block").
//
// ViTPostEncoder / GPTPostEncoder read ONE row per item (cvap/module/val.py:288-289, 143-145), so the last block's attention has one
// query per (item, head).  Round 3 still projected every token to K and V (a [M, 2D] contraction, its dX and its dW: 1.0 ms per step)
// for those queries to look at.  With one query the projections fold into the query side:
//     scores_j   = q_h . (W_k,h h1_j + b_k,h) / 8 = (W_k,h^T q_h) . h1_j / 8 + const          (the constant cancels in the softmax)
//     o_h        = sum_j p_j (W_v,h h1_j + b_v,h)  = W_v,h (sum_j p_j h1_j) + b_v,h            (sum_j p_j = 1)
// i.e. per (item, head) a query-like vector qk_h = W_k,h^T q_h of width D and a context ctx_h = sum_j p_j h1_j of width D, both
// against the LayerNorm output h1 itself.  The projections that remain are per-head products on the `batch` read-out rows
// (`vipant_gemm_nt_heads`; the block-sparse [batch * H, D] operand of `vipant_head_expand` only feeds the two weight gradients).
// Backward: with dctx_h = W_v,h^T do_h,
//     dp_j = dctx_h . h1_j,  delta = dctx_h . ctx_h,  ds_j = p_j (dp_j - delta) / 8,
//     dh1_j = sum_h p_j,h dctx_h + ds_j,h qk_h        (rank-2H update per token),     dqk_h = sum_j ds_j,h h1_j.
// Both kernels stream h1 (fwd: read; bwd: read + write dh1) in place of the K / V projection, its dX, its dW and the two one-query
// kernels.  One workgroup of four waves per item, rounds of 32 tokens staged in LDS by LDS-DMA; every product -- the dot products over
// D (scores, dp), the two sums over tokens (contexts, dqk: transposed fragments of the staged rows) and the rank-2H update that is
// dh1 -- runs on v_mfma_f32_16x16x32_bf16 with the heads padded to 16; the softmax statistics are taken one lane per token.
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
__device__ __forceinline__ int key_limit(const int64_t* idx, int item, int S, int causal) {
    if (!causal) return S;
    if (idx == nullptr) return 1;          // causal, read-out row 0: one key (as csrc/readout_rows.hip's row_index gives)
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

// ---- LDS images of token rows (the attention kernels' format, csrc/attention.hip): [32 rows x 64 columns] bf16, 128-byte rows, the
// 16-byte chunk c of row r stored at position c ^ img_swz(r).  A round's 32 token rows of h1 are D / 64 such images side by side.
// Row fragments (A operand rows = tokens, K = the image's columns) are plain 16-byte reads; transposed fragments (operand rows = the
// image's COLUMNS, K = its 32 token rows) come from ds_read_b64_tr_b16: both conflict-free under this swizzle.
__device__ __forceinline__ int img_swz(int r) { return ((r >> 1) & 3) << 1; }
struct ImgLane {
    uint32_t row[2];   // row fragment of 16-row tile 0, column step 0 / 1 (32 columns each)
    uint32_t tr[4];    // transposed fragment, first read, column tile 0..3 (16 columns each)
};
__device__ __forceinline__ ImgLane img_lane(int lane) {
    ImgLane a;
    const int r = lane & 15, g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) a.row[ds] = (uint32_t)(r * 128 + (((ds * 4 + g) ^ img_swz(r)) << 4));
    const int ra = 4 * g + qq;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) a.tr[dt] = (uint32_t)(ra * 128 + (((2 * dt + (pp >> 1)) ^ img_swz(ra)) << 4) + (pp & 1) * 8);
    return a;
}
__device__ __forceinline__ bf16x8 img_row_frag(const char* img, const ImgLane& a, int tile, int ds) {
    return *(const bf16x8*)(img + a.row[ds] + tile * 2048);
}
// operand[row = column 16 dt + (lane & 15) of the image][k-slot (g, e)]: e < 4 -> image row 4 g + e, e >= 4 -> image row 16 + 4 g + (e - 4)
__device__ __forceinline__ bf16x8 img_tr_frag(const char* img, const ImgLane& a, int dt) {
    const bf16x4 lo = lds_read_tr16(img + a.tr[dt]);
    const bf16x4 hi = lds_read_tr16(img + a.tr[dt] + 2048);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the other operand of such a product, from a [16 heads][32 tokens] bf16 array: the same token order
__device__ __forceinline__ bf16x8 head_token_frag(const bf16_t* buf, int r, int g) {
    const bf16x4 lo = *(const bf16x4*)(buf + r * 32 + 4 * g);
    const bf16x4 hi = *(const bf16x4*)(buf + r * 32 + 16 + 4 * g);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// rows j0 .. j0 + 31 of an item (rows past the end read as zeros through the buffer descriptor) -> NI images, by LDS-DMA
template <int NI>
__device__ __forceinline__ void fill_images(char* imgs, __amdgpu_buffer_rsrc_t rs, int j0, int wave, int lane) {
    for (int q = wave; q < NI * 4; q += 4) {
        const int img = q >> 2, blk = q & 3;
        const int r = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ img_swz(r);
        lds_dma16(rs, imgs + img * 4096 + blk * 1024, (uint32_t)(j0 + r) * (NI * 128) + (uint32_t)(img * 128 + c * 16), 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// (A) of both kernels: [32 tokens] x [16 heads] dot products over D: wave w takes token tile w & 1 and the K half w >> 1, the two
// halves are added by the reader.  part: [2 halves][16 heads][32 tokens] fp32; bq: the heads' vectors as B fragments, this wave's K half
// (a register array indexed by the wave's number would live in scratch memory).
// PAIR: the heads' vectors are bf16 pairs (hi + lo, round 5): the lo plane's fragments take a second MFMA against the same token rows.
template <int NI, bool PAIR>
__device__ __forceinline__ void dots_round(const char* imgs, const ImgLane& a, const bf16x8 (&bq)[NI], const bf16x8 (&bl)[PAIR ? NI : 1],
                                           float* part, int wave, int r, int g) {
    const int tile = wave & 1, kh = wave >> 1;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ii = 0; ii < NI / 2; ++ii) {
        const int img = kh * (NI / 2) + ii;
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
            const bf16x8 fr = img_row_frag(imgs + img * 4096, a, tile, ds);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, bq[2 * ii + ds], acc, 0, 0, 0);
            if (PAIR) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr, bl[2 * ii + ds], acc, 0, 0, 0);
        }
    }
    *(f32x4*)(part + (kh * 16 + r) * 32 + 16 * tile + 4 * g) = acc;
}

// ctx[item, h, :] = sum_j softmax_j(qk[item, h, :] . h1[item, j, :] / 8) h1[item, j, :];  probs[item, h, j] = that softmax (fp32).
// One workgroup per item, rounds of 32 tokens staged in LDS as images: (A) the scores of the round against ALL heads (padded to 16)
// on the matrix pipe; (B) wave w owns heads HPW w ..: one lane per token takes the round's maximum, the rescale factor and the
// exponentials, left in LDS as bf16 P[head][token]; (C) ctx^T[columns x heads] += h1^T[columns x tokens] . P^T[tokens x heads], wave w
// owning columns D / 4 * w ..: transposed fragments of the images.  Raw scores wait in LDS until maximum and sum are final.
// PAIR (round 5): qk and ctx are bf16 pairs, x = hi + lo in two planes `lo` elements apart -- these [batch * H, D] tensors carried a bf16
// rounding per (item, head) that the full block does not have (profiles/r5_parity_observed.jsonl); the kernel is HBM-bound on h1.
template <int NI, int HPW, int MINB, bool PAIR>
__global__ __launch_bounds__(256, MINB) void rows_ctx_fwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ h1,
                                                                 const int64_t* __restrict__ idx, bf16_t* __restrict__ ctx,
                                                                 float* __restrict__ probs, int S, int causal, int64_t lo) {
    constexpr int D = NI * 64, NH = 4 * HPW, IPW = NI / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* imgs = smem;                                          // NI x 4 KiB
    float* part = (float*)(smem + NI * 4096);                   // [2][16][32] fp32
    bf16_t* pbuf = (bf16_t*)(part + 2 * 16 * 32);               // [16][32] bf16
    float* fbuf = (float*)(pbuf + 16 * 32);                     // [16] rescale factors of the round, then 1 / sum
    float* sc = fbuf + 16;                                      // [NH][Sp] raw scores (exp2 domain)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const int item = blockIdx.x;
    const int nkeys = key_limit(idx, item, S, causal);
    const int Sp = (S + 31) & ~31;
    const int64_t head0 = (int64_t)item * NH + HPW * wave;
    const bf16_t* rows = h1 + (int64_t)item * S * D;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(rows, (uint32_t)S * D * 2);
    const ImgLane a = img_lane(lane);
    bf16x8 bq[NI], bl[PAIR ? NI : 1];
    {
        const bf16_t* qrow = qk + ((int64_t)item * NH + (r < NH ? r : 0)) * D + (wave >> 1) * (D / 2) + 8 * g;
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            bq[s] = *(const bf16x8*)(qrow + 32 * s);
            if (PAIR) bl[s] = *(const bf16x8*)(qrow + lo + 32 * s);
            if (r >= NH) { bq[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; if (PAIR) bl[s] = bq[s]; }
        }
    }
    for (int i = threadIdx.x; i < 16 * 32 / 2; i += 256) ((uint32_t*)pbuf)[i] = 0u;        // heads NH .. 15 stay zero
    if (threadIdx.x < 16) fbuf[threadIdx.x] = 0.f;
    f32x4 acc[IPW][4];
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[ii][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[HPW], lsum[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) { m[h] = -INFINITY; lsum[h] = 0.f; }
    for (int j0 = 0; j0 < nkeys; j0 += 32) {
        fill_images<NI>(imgs, rs, j0, wave, lane);
        __syncthreads();
        dots_round<NI, PAIR>(imgs, a, bq, bl, part, wave, r, g);                              // (A)
        __syncthreads();
#pragma unroll
        for (int h = 0; h < HPW; ++h) {                                                       // (B)
            const int hg = HPW * wave + h, j = j0 + lane;
            const bool on = lane < 32 && j < nkeys;
            const float sv = on ? (part[hg * 32 + (lane & 31)] + part[(16 + hg) * 32 + (lane & 31)]) * (0.125f * LOG2E_) : -INFINITY;
            const float mn = fmaxf(m[h], wave_max_dpp(sv));
            const float f = __builtin_amdgcn_exp2f(m[h] - mn);
            const float pj = __builtin_amdgcn_exp2f(sv - mn);
            lsum[h] = lsum[h] * f + pj;                 // per-lane share of the sum; added up across lanes at the end
            m[h] = mn;
            if (lane < 32) {
                pbuf[hg * 32 + lane] = (bf16_t)pj;
                sc[hg * Sp + j] = sv;
            }
            if (lane == 0) fbuf[hg] = f;
        }
        __syncthreads();
        {                                                                                     // (C)
            const float f = fbuf[r];
            const bf16x8 bp = head_token_frag(pbuf, r, g);
#pragma unroll
            for (int ii = 0; ii < IPW; ++ii)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    acc[ii][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag(imgs + (IPW * wave + ii) * 4096, a, dt), bp,
                                                                          acc[ii][dt] * f, 0, 0, 0);
        }
        __syncthreads();
    }
    float inv[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        inv[h] = 1.0f / wave_sum_dpp(lsum[h]);
        if (lane == 0) fbuf[HPW * wave + h] = inv[h];
    }
    __syncthreads();
    if (r < NH) {       // lane: head r, columns 64 img + 16 dt + 4 g .. + 3
        const float iv = fbuf[r];
        bf16_t* orow = ctx + ((int64_t)item * NH + r) * D + 4 * g;
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = acc[ii][dt] * iv;
                const bf16x4 hi = f32x4_to_bf16x4(v);
                *(bf16x4*)(orow + 64 * (IPW * wave + ii) + 16 * dt) = hi;
                if (PAIR)
                    *(bf16x4*)(orow + lo + 64 * (IPW * wave + ii) + 16 * dt) =
                        f32x4_to_bf16x4(f32x4{v[0] - (float)hi[0], v[1] - (float)hi[1], v[2] - (float)hi[2], v[3] - (float)hi[3]});
            }
    }
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        float* pout = probs + (head0 + h) * S;
        for (int j = lane; j < S; j += 64)
            pout[j] = j < nkeys ? __builtin_amdgcn_exp2f(sc[(HPW * wave + h) * Sp + j] - m[h]) * inv[h] : 0.f;
    }
}

// dh1[item, j, :] = sum_h p_j,h dctx_h + ds_j,h qk_h;  dqk[item, h, :] = sum_j ds_j,h h1_j;  ds_j,h = p_j,h (dctx_h . h1_j - dctx_h . ctx_h) / 8.
// Rounds of 32 tokens as in the forward: (A) dctx_h . h1_j on the matrix pipe; (B) one lane per token: ds for the wave's heads, left in
// LDS as bf16 dS[head][token] (for C) and, with p, as the token's row [p_0 .. p_15 | ds_0 .. ds_15] (for D); (C) dqk^T[columns x heads]
// += h1^T . dS^T from transposed image fragments, wave w owning columns D / 4 * w ..; (D) dh1 of the round as ONE product per 16 tokens
// x 16 columns: (dh1 tile)^T = [dctx | qk]^T [16 columns x 32] . [p | ds]^T [32 x 16 tokens] -- the left operand, for the wave's D / 4
// columns, is resident in registers (built once per item through the image area), the result leaves as 16-byte stores.
// PAIR (round 5): qk, ctx and dqk are bf16 pairs (planes `lo` elements apart); dctx is a single bf16 plane.  ctx enters delta = dctx . ctx
// with both planes, dqk leaves as a pair (its consumer, dq = W_k dqk, takes both planes).  dctx = W_v^T do is a linear image of `do`,
// which is bf16 itself, and enters dp = dctx . h1 and delta with the SAME rounded value, so its rounding is a 2^-9 relative
// perturbation of `do` that the difference dp - delta does not amplify; a second plane for it would cost the 48 registers of a second
// resident fragment set (the kernel spills at 256).  The resident [dctx | qk]^T operand of (D) keeps qk's hi plane: dh1 leaves as one
// bf16 per element anyway.
template <int NI, int HPW, int MINB, bool PAIR>
__global__ __launch_bounds__(256, MINB) void rows_ctx_bwd_kernel(const bf16_t* __restrict__ qk, const bf16_t* __restrict__ dctx,
                                                                 const bf16_t* __restrict__ ctx, const bf16_t* __restrict__ h1,
                                                                 const int64_t* __restrict__ idx, const float* __restrict__ probs,
                                                                 bf16_t* __restrict__ dh1, bf16_t* __restrict__ dqk, int S, int causal,
                                                                 int64_t lo, float* __restrict__ dbk) {
    constexpr int D = NI * 64, NH = 4 * HPW, IPW = NI / 4, CT = NI, EPL = NI;     // CT: 16-column tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* imgs = smem;                                          // NI x 4 KiB (first: [D][32] bf16, the item's [dctx | qk]^T)
    float* part = (float*)(smem + NI * 4096);                   // [2][16][32] fp32
    bf16_t* dsbuf = (bf16_t*)(part + 2 * 16 * 32);              // [16][32] bf16: dS[head][token]
    bf16_t* pds = dsbuf + 16 * 32;                              // [32 tokens][32]: p of 16 heads | ds of 16 heads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, r = lane & 15;
    const int item = blockIdx.x;
    // d b_k = 0 EXACTLY (a bias on the keys adds the same constant to every score of a query: the softmax does not see it); written
    // here, by the first workgroup, so that the caller needs no fill launch for it
    if (item == 0 && dbk != nullptr)
        for (int i = threadIdx.x; i < D; i += 256) dbk[i] = 0.f;
    const int nkeys = key_limit(idx, item, S, causal);
    const int64_t head0 = (int64_t)item * NH + HPW * wave;
    const bf16_t* rows = h1 + (int64_t)item * S * D;
    bf16_t* drows = dh1 + (int64_t)item * S * D;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(rows, (uint32_t)S * D * 2);
    const ImgLane a = img_lane(lane);
    // the item's [dctx | qk]^T through the image area into registers; zero padding of the small arrays
    bf16_t* dcq = (bf16_t*)imgs;
    for (int i = threadIdx.x; i < (D * 32 + 2 * 16 * 32 * 2 + 16 * 32 + 32 * 32) / 8; i += 256) ((u32x4*)smem)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int i = threadIdx.x; i < NH * (D / 8); i += 256) {      // 8 columns of one head per thread
        const int hh = i / (D / 8), n8 = (i - hh * (D / 8)) * 8;
        const bf16x8 vd = *(const bf16x8*)(dctx + ((int64_t)item * NH + hh) * D + n8);
        const bf16x8 vq = *(const bf16x8*)(qk + ((int64_t)item * NH + hh) * D + n8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = n8 + e, sw = (n >> 2) & 3;
            dcq[n * 32 + 8 * ((hh >> 3) ^ sw) + (hh & 7)] = vd[e];
            dcq[n * 32 + 8 * ((2 + (hh >> 3)) ^ sw) + (hh & 7)] = vq[e];
        }
    }
    bf16x8 bq[NI], bl[1];
    {
        const bf16_t* qrow = dctx + ((int64_t)item * NH + (r < NH ? r : 0)) * D + (wave >> 1) * (D / 2) + 8 * g;
#pragma unroll
        for (int s = 0; s < NI; ++s) {
            bq[s] = *(const bf16x8*)(qrow + 32 * s);
            if (r >= NH) bq[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    float delta[HPW];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        float dc[EPL], c[EPL];
        load_row<EPL>(dctx + (head0 + h) * D, lane, dc);
        load_row<EPL>(ctx + (head0 + h) * D, lane, c);
        if (PAIR) {
            float cl[EPL];
            load_row<EPL>(ctx + lo + (head0 + h) * D, lane, cl);
#pragma unroll
            for (int e = 0; e < EPL; ++e) c[e] += cl[e];
        }
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) d = __builtin_fmaf(dc[e], c[e], d);
        delta[h] = wave_sum_dpp(d);
    }
    __syncthreads();
    // operand row m = 4 g' + i' of the product (pair p, half c) is column 32 p + 8 g' + 4 c + i' of the wave's share: a lane's results
    // of the two halves are then 8 consecutive columns of its token, one 16-byte store
    bf16x8 aq[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int n = 16 * CT * wave + 32 * (ct >> 1) + 8 * (r >> 2) + 4 * (ct & 1) + (r & 3);
        aq[ct] = *(const bf16x8*)(dcq + n * 32 + 8 * (g ^ ((n >> 2) & 3)));
    }
    f32x4 acc[IPW][4];
#pragma unroll
    for (int ii = 0; ii < IPW; ++ii)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[ii][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int j0 = 0; j0 < S; j0 += 32) {
        const bool live = j0 < nkeys;                   // rounds behind a causal limit: zeros for dh1, nothing else
        float pj[HPW];
#pragma unroll
        for (int h = 0; h < HPW; ++h) pj[h] = (lane < 32 && j0 + lane < nkeys) ? probs[(head0 + h) * S + j0 + lane] : 0.f;
        if (live) fill_images<NI>(imgs, rs, j0, wave, lane);
        __syncthreads();
        if (live) dots_round<NI, false>(imgs, a, bq, bl, part, wave, r, g);                     // (A)
        __syncthreads();
        if (lane < 32) {                                                                      // (B)
            const int sw = (lane >> 2) & 3;
#pragma unroll
            for (int h = 0; h < HPW; ++h) {
                const int hg = HPW * wave + h;
                const float ds = live ? pj[h] * (part[hg * 32 + lane] + part[(16 + hg) * 32 + lane] - delta[h]) * 0.125f : 0.f;
                dsbuf[hg * 32 + lane] = (bf16_t)ds;
                pds[lane * 32 + 8 * ((hg >> 3) ^ sw) + (hg & 7)] = (bf16_t)pj[h];
                pds[lane * 32 + 8 * ((2 + (hg >> 3)) ^ sw) + (hg & 7)] = (bf16_t)ds;
            }
        }
        __syncthreads();
        if (live) {                                                                           // (C)
            const bf16x8 bd = head_token_frag(dsbuf, r, g);
#pragma unroll
            for (int ii = 0; ii < IPW; ++ii)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    acc[ii][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag(imgs + (IPW * wave + ii) * 4096, a, dt), bd,
                                                                          acc[ii][dt], 0, 0, 0);
        }
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {                                                // (D)
            const int tl = 16 * tile + r, tok = j0 + tl;
            const bf16x8 bt = *(const bf16x8*)(pds + tl * 32 + 8 * (g ^ ((tl >> 2) & 3)));
            bf16_t* orow = drows + (int64_t)tok * D + 16 * CT * wave + 8 * g;
#pragma unroll
            for (int pr = 0; pr < CT / 2; ++pr) {
                const f32x4 c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[2 * pr], bt, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4 c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[2 * pr + 1], bt, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const bf16x4 lo = f32x4_to_bf16x4(c0), hi = f32x4_to_bf16x4(c1);
                if (tok < S) *(bf16x8*)(orow + 32 * pr) = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
        __syncthreads();
    }
    if (r < NH) {
        bf16_t* orow = dqk + ((int64_t)item * NH + r) * D + 4 * g;
#pragma unroll
        for (int ii = 0; ii < IPW; ++ii)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 v = acc[ii][dt];
                const bf16x4 hi = f32x4_to_bf16x4(v);
                *(bf16x4*)(orow + 64 * (IPW * wave + ii) + 16 * dt) = hi;
                if (PAIR)
                    *(bf16x4*)(orow + lo + 64 * (IPW * wave + ii) + 16 * dt) =
                        f32x4_to_bf16x4(f32x4{v[0] - (float)hi[0], v[1] - (float)hi[1], v[2] - (float)hi[2], v[3] - (float)hi[3]});
            }
    }
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
    VIPANT_REQUIRE(batch > 0 && S > 0 && S <= 1024 && (H == 8 || H == 12 || H == 16), VIPANT_EBADSHAPE,
                   "rows_ctx: batch=%ld S=%ld H=%ld (heads of 64: H = 8, 12 or 16, i.e. width 512, 768 or 1024; S <= 1024: the raw scores of an item wait in LDS)", (long)batch,
                   (long)S, (long)H);
    return VIPANT_OK;
}

template <int EPL, int HPW, int MINB, bool PAIR>
int32_t launch_fwd(const bf16_t* qk, const bf16_t* h1, const int64_t* idx, bf16_t* ctx, float* probs, int batch, int S, int causal,
                   int64_t lo, hipStream_t st) {
    const int lds = EPL * 4096 + 4096 + 1024 + 64 + 4 * HPW * ((S + 31) & ~31) * (int)sizeof(float);
    static DeviceMax most;
    if (raise_on_device(most, lds)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)rows_ctx_fwd_kernel<EPL, HPW, MINB, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        raised_on_device(most, lds);
    }
    hipLaunchKernelGGL((rows_ctx_fwd_kernel<EPL, HPW, MINB, PAIR>), dim3((unsigned)batch), dim3(256), lds, st, qk, h1, idx, ctx, probs, S, causal,
                       lo);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int EPL, int HPW, int MINB, bool PAIR>
int32_t launch_bwd(const bf16_t* qk, const bf16_t* dctx, const bf16_t* ctx, const bf16_t* h1, const int64_t* idx, const float* probs,
                   bf16_t* dh1, bf16_t* dqk, int batch, int S, int causal, int64_t lo, float* dbk, hipStream_t st) {
    constexpr int lds = EPL * 4096 + 4096 + 1024 + 2048;
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)rows_ctx_bwd_kernel<EPL, HPW, MINB, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done_on_device(once);
    }
    hipLaunchKernelGGL((rows_ctx_bwd_kernel<EPL, HPW, MINB, PAIR>), dim3((unsigned)batch), dim3(256), lds, st, qk, dctx, ctx, h1, idx, probs, dh1,
                       dqk, S, causal, lo, dbk);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

extern "C" int32_t vipant_rows_ctx_fwd(const uint16_t* qk, const uint16_t* h1, const int64_t* idx, uint16_t* ctx, float* probs,
                                       int64_t batch, int64_t S, int64_t H, int32_t causal, int32_t pair, void* stream) {
    if (int32_t e = check_ctx(batch, S, H)) return e;
    VIPANT_REQUIRE(qk != nullptr && h1 != nullptr && ctx != nullptr && probs != nullptr, VIPANT_EBADSHAPE, "rows_ctx_fwd: null operand");
    VIPANT_REQUIRE((uintptr_t)qk % 16 == 0 && (uintptr_t)h1 % 16 == 0 && (uintptr_t)ctx % 16 == 0, VIPANT_EALIGN,
                   "rows_ctx_fwd: operands must be 16-byte aligned");
    const bf16_t *a = (const bf16_t*)qk, *b = (const bf16_t*)h1;
    const int64_t lo = batch * H * H * 64;          // the lo planes of qk / ctx: [2][batch * H][D]
    hipStream_t st = (hipStream_t)stream;
    if (pair) {
        if (H == 12) return launch_fwd<12, 3, 2, true>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, lo, st);
        if (H == 16) return launch_fwd<16, 4, 1, true>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, lo, st);
        return launch_fwd<8, 2, 2, true>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, lo, st);
    }
    if (H == 12) return launch_fwd<12, 3, 2, false>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, 0, st);
    if (H == 16) return launch_fwd<16, 4, 1, false>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, 0, st);
    return launch_fwd<8, 2, 2, false>(a, b, idx, (bf16_t*)ctx, probs, (int)batch, (int)S, causal, 0, st);
}

extern "C" int32_t vipant_rows_ctx_bwd(const uint16_t* qk, const uint16_t* dctx, const uint16_t* ctx, const uint16_t* h1,
                                       const int64_t* idx, const float* probs, uint16_t* dh1, uint16_t* dqk, float* dbk, int64_t batch,
                                       int64_t S, int64_t H, int32_t causal, int32_t pair, void* stream) {
    if (int32_t e = check_ctx(batch, S, H)) return e;
    VIPANT_REQUIRE(qk != nullptr && dctx != nullptr && ctx != nullptr && h1 != nullptr && probs != nullptr && dh1 != nullptr && dqk != nullptr,
                   VIPANT_EBADSHAPE, "rows_ctx_bwd: null operand");
    VIPANT_REQUIRE((uintptr_t)qk % 16 == 0 && (uintptr_t)dctx % 16 == 0 && (uintptr_t)ctx % 16 == 0 && (uintptr_t)h1 % 16 == 0 &&
                   (uintptr_t)dh1 % 16 == 0 && (uintptr_t)dqk % 16 == 0, VIPANT_EALIGN, "rows_ctx_bwd: operands must be 16-byte aligned");
    const bf16_t *a = (const bf16_t*)qk, *b = (const bf16_t*)dctx, *c = (const bf16_t*)ctx, *d = (const bf16_t*)h1;
    hipStream_t st = (hipStream_t)stream;
    const int64_t lo = batch * H * H * 64;
    if (pair) {
        if (H == 12) return launch_bwd<12, 3, 2, true>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, lo, dbk, st);
        if (H == 16) return launch_bwd<16, 4, 1, true>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, lo, dbk, st);
        return launch_bwd<8, 2, 2, true>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, lo, dbk, st);
    }
    if (H == 12) return launch_bwd<12, 3, 2, false>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, 0, dbk, st);
    if (H == 16) return launch_bwd<16, 4, 1, false>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, 0, dbk, st);
    return launch_bwd<8, 2, 2, false>(a, b, c, d, idx, probs, (bf16_t*)dh1, (bf16_t*)dqk, (int)batch, (int)S, causal, 0, dbk, st);
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
