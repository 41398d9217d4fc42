// Multi-head attention core, head dim 64: softmax(q k^T / 8 [+ causal]) v and its backward
// (nn.MultiheadAttention inside ResidualAttentionBlock, cvap/module/val.py:511-517).
//
// gfx950 design.  Sequences on this path are short (S <= 316 audio tokens, <= 77 text tokens), so one
// head's whole K and V (<= 40 KiB each in bf16) live in LDS: one workgroup per (batch, head), no online
// softmax, no second pass over keys (S > 384 takes the streaming variants at the end of the file).  All products run on v_mfma_f32_16x16x32_bf16 with the QUERY on the
// MFMA column (lane & 15):
//     S^T tile  = K_tile . Q^T          (A = K rows from LDS by ds_read_b128, B = Q rows from registers)
//     O^T tile  = V^T . P^T             (A = V^T by ds_read_b64_tr_b16 transposed reads, B = P in place)
// The S^T accumulator (keys on registers, query on the lane) is, after exp and bf16 packing, already the
// B operand of the second product -- no LDS round trip, no cross-lane movement; the k-slot order it implies
// (slot j<4 -> key tile 2u, j>=4 -> key tile 2u+1) is matched by which rows the transposed reads fetch.
// Row max / sum need only two xor-shuffles (lanes l, l^16, l^32, l^48 share a query).
// One LDS image per matrix serves both row reads and transposed reads: 128-B rows, 16-B chunk index
// XOR-ed with ((row>>1)&3)<<1 (conflict-free for both access kinds); filled by LDS-DMA with the swizzle
// applied to the source address.  Backward = two passes with the same structure (dQ per query block with
// K,V resident; dK,dV per key block with Q,dO resident): 40 % more MFMA work than a single-pass scheme but
// no float atomics, bitwise reproducible.
#include <stdlib.h>

#include "attn_common.h"

using namespace vipant_attn;

namespace {

__device__ __forceinline__ int img_swz(int r) { return ((r >> 1) & 3) << 1; }

// Fill a [rows8*8 x 64] bf16 LDS image from `rows8*8` consecutive rows (stride ld_bytes) of a buffer.
__device__ __forceinline__ void dma_image(char* lds, __amdgpu_buffer_rsrc_t rs, uint32_t ld_bytes, int rows8,
                                          int wave, int nwaves, int lane) {
    for (int blk = wave; blk < rows8; blk += nwaves) {
        const int r = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ img_swz(r);
        lds_dma16(rs, lds + blk * 1024, (uint32_t)r * ld_bytes + (uint32_t)c * 16, 0);
    }
}

// Per-lane byte offsets into an image; everything else is a compile-time constant added on top, because
// the swizzle term depends only on (row & 7) and tile / k-step bases are multiples of 16 rows.
struct ImgLane {
    uint32_t row[2];   // row-read fragment of 16-row tile 0 for d-step 0 / 1
    uint32_t tr[4];    // transposed fragment, first read, rows 4g+qq of tile 0, d-tile 0..3
};
__device__ __forceinline__ ImgLane img_lane(int lane) {
    ImgLane a;
    const int r = lane & 15, g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) a.row[ds] = (uint32_t)(r * 128 + (((ds * 4 + g) ^ img_swz(r)) << 4));
    const int ra = 4 * g + qq;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
        a.tr[dt] = (uint32_t)(ra * 128 + (((2 * dt + (pp >> 1)) ^ img_swz(ra)) << 4) + (pp & 1) * 8);
    return a;
}

// Row-read fragment (A operand rows / B operand columns): 16 B = row (tile*16 + (lane&15)), k = 32*ds + 8*(lane>>4)..
__device__ __forceinline__ bf16x8 img_row_frag(const char* img, const ImgLane& a, int tile, int ds) {
    return *(const bf16x8*)(img + a.row[ds] + tile * 2048);
}

// Transposed fragment: A[row = 16*dt + (lane&15)][k-slot (g, j)] with slot j<4 -> image row 32u + 4g + j,
// j>=4 -> image row 32u + 16 + 4g + (j-4).
__device__ __forceinline__ bf16x8 img_tr_frag(const char* img, const ImgLane& a, int u, int dt) {
    const bf16x4 lo = lds_read_tr16(img + a.tr[dt] + u * 4096);
    const bf16x4 hi = lds_read_tr16(img + a.tr[dt] + u * 4096 + 2048);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 r;
    r[0] = (bf16_t)a[0]; r[1] = (bf16_t)a[1]; r[2] = (bf16_t)a[2]; r[3] = (bf16_t)a[3];
    r[4] = (bf16_t)b[0]; r[5] = (bf16_t)b[1]; r[6] = (bf16_t)b[2]; r[7] = (bf16_t)b[3];
    return r;
}

// the same through four explicit pair conversions (element inserts into a bf16x8 made hipcc convert single values and merge
// them with v_perm_b32 when the elements became available one at a time)
__device__ __forceinline__ bf16x8 pack8_pairs(f32x4 a, f32x4 b) {
    bf16x2 p0, p1, p2, p3;
    p0[0] = (bf16_t)a[0]; p0[1] = (bf16_t)a[1]; p1[0] = (bf16_t)a[2]; p1[1] = (bf16_t)a[3];
    p2[0] = (bf16_t)b[0]; p2[1] = (bf16_t)b[1]; p3[0] = (bf16_t)b[2]; p3[1] = (bf16_t)b[3];
    u32x4 r;
    r[0] = __builtin_bit_cast(uint32_t, p0); r[1] = __builtin_bit_cast(uint32_t, p1);
    r[2] = __builtin_bit_cast(uint32_t, p2); r[3] = __builtin_bit_cast(uint32_t, p3);
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lanes sharing (lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Make the compiler wait for a prefetched fragment HERE (before the block's output stores are issued): vmcnt counts loads and
// stores in one in-order counter, so a wait placed after the stores would also wait for their write acknowledgements.
__device__ __forceinline__ void settle(bf16x8& f) { lds_raw_use(f); }

template <int V> struct Int2 { static constexpr int value = V; };


// ------------------------------------------------------------------------------------------- forward
template <int NT, bool CAUSAL, int NW, int EDGE>
__global__ __launch_bounds__(NW * 64, 2) void mha_fwd_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NT * 16;
    char* kimg = smem;
    char* vimg = smem + SP * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;

    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    dma_image(kimg, uniform_rsrc(base + D, lim > (uint32_t)(D * 2) ? lim - D * 2 : 0), ld * 2, SP / 8, wave, NW, lane);
    dma_image(vimg, uniform_rsrc(base + 2 * D, lim > (uint32_t)(D * 4) ? lim - D * 4 : 0), ld * 2, SP / 8, wave, NW, lane);
    __syncthreads();

    const int qcol = lane & 15, g = lane >> 4;
    const ImgLane il = img_lane(lane);
    // The query fragments of a block come straight from HBM: fetch block i+1's while block i computes (the first
    // fetch flies under the K/V DMA), otherwise every block starts with an exposed global-load latency.
    auto load_q = [&](int qb, bf16x8& f0, bf16x8& f1) {
        const int qq = qb * 16 + qcol;
        const bf16_t* qp = base + (int64_t)(qq < p.S ? qq : p.S - 1) * ld + 8 * g;
        f0 = *(const bf16x8*)qp;
        f1 = *(const bf16x8*)(qp + 32);
    };
    bf16x8 qn0, qn1;
    load_q(wave, qn0, qn1);
    settle(qn0); settle(qn1);           // so that the loop header never carries a vmcnt wait (it would also cover the stores)
    for (int qb = wave; qb * 16 < p.S; qb += NW) {
        const int q = qb * 16 + qcol;
        const bf16x8 qf0 = qn0, qf1 = qn1;
        if ((qb + NW) * 16 < p.S) load_q(qb + NW, qn0, qn1);

        f32x4 s[NT];
        float m = -INFINITY;
        // Fragment pipeline (as in the contractions): the K row fragments of key-tile pair u + 2 are requested before the
        // MFMAs of pair u, through a 3-deep register ring, so an LDS round trip is never exposed in front of an MFMA group.
        bf16x8 kr[3][4];
        auto k_pair = [&](int u, bf16x8 (&f)[4]) {
            f[0] = img_row_frag(kimg, il, 2 * u, 0); f[1] = img_row_frag(kimg, il, 2 * u, 1);
            f[2] = img_row_frag(kimg, il, 2 * u + 1, 0); f[3] = img_row_frag(kimg, il, 2 * u + 1, 1);
        };
        k_pair(0, kr[0]);
        if (NT > 2) k_pair(1, kr[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            if (u + 2 < NT / 2) k_pair(u + 2, kr[(u + 2) % 3]);
            f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[u % 3][0], qf0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[u % 3][2], qf0, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[u % 3][1], qf1, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[u % 3][3], qf1, a1, 0, 0, 0);
            // masking code exists only for the last EDGE key tiles (compile-time): with NT chosen as the smallest
            // even tile count that covers S, at most the last two tiles can hold keys >= S; the causal (text) variant
            // masks everywhere.
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int kt = 2 * u + h2;
                f32x4 acc = h2 ? a1 : a0;
                if (CAUSAL || kt >= NT - EDGE) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 16 + g * 4 + r;
                        if (key >= p.S || (CAUSAL && key > q)) acc[r] = -INFINITY;
                    }
                }
                m = fmaxf(fmaxf(m, acc[0]), acc[1]);          // two v_max3_f32 per tile
                m = fmaxf(fmaxf(m, acc[2]), acc[3]);
                s[kt] = acc;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        m = group_max(m);
        // the first two V^T fragment groups fly under the exponentials
        bf16x8 vr[3][4];
        auto v_grp = [&](int u, bf16x8 (&f)[4]) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) f[dt] = img_tr_frag(vimg, il, u, dt);
        };
        v_grp(0, vr[0]);
        if (NT > 2) v_grp(1, vr[1]);
        float l = 0.f;
        const float mc = m * C2;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // exponentials and the second product are fused per 32-key step: the MFMAs of step u run on the matrix pipe while
        // the VALU works on the exponentials of step u + 1
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            if (u + 2 < NT / 2) v_grp(u + 2, vr[(u + 2) % 3]);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[2 * u + t][r] * C2 - mc);
                    s[2 * u + t][r] = e;
                    l += e;
                }
            const bf16x8 pf = pack8(s[2 * u], s[2 * u + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[u % 3][dt], pf, o[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        l = group_sum(l);
        settle(qn0); settle(qn1);          // next block's query fragments have arrived; the stores below drain under its MFMAs
        if (q < p.S) {
            const float inv = __frcp_rn(l);
            bf16_t* op = p.out + (row_base + q) * D + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(op + dt * 16) = f32x4_to_bf16x4(o[dt] * inv);
            if (g == 0) p.lse[((int64_t)b * p.H + h) * p.S + q] = m * SCALE + __logf(l);
        }
    }
}


// ---------------------------------------------------------------------------- backward, pass A: dQ
template <int NT, bool CAUSAL, int NW, int EDGE>
__global__ __launch_bounds__(NW * 64, 2) void mha_bwd_dq_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NT * 16;
    char* kimg = smem;
    char* vimg = smem + SP * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;

    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    dma_image(kimg, uniform_rsrc(base + D, lim > (uint32_t)(D * 2) ? lim - D * 2 : 0), ld * 2, SP / 8, wave, NW, lane);
    dma_image(vimg, uniform_rsrc(base + 2 * D, lim > (uint32_t)(D * 4) ? lim - D * 4 : 0), ld * 2, SP / 8, wave, NW, lane);
    __syncthreads();

    const int qcol = lane & 15, g = lane >> 4;
    const ImgLane il = img_lane(lane);
    struct QBlock { bf16x8 q0, q1, do0, do1, o0, o1; float lse; };
    auto load_blk = [&](int qb, QBlock& t) {
        const int qq = qb * 16 + qcol;
        const int qrow = qq < p.S ? qq : p.S - 1;
        const bf16_t* qp = base + (int64_t)qrow * ld + 8 * g;
        t.q0 = *(const bf16x8*)qp;
        t.q1 = *(const bf16x8*)(qp + 32);
        const int64_t orow = (row_base + qrow) * D + h * 64 + 8 * g;
        t.do0 = *(const bf16x8*)(p.dout + orow);
        t.do1 = *(const bf16x8*)(p.dout + orow + 32);
        t.o0 = *(const bf16x8*)(p.out + orow);
        t.o1 = *(const bf16x8*)(p.out + orow + 32);
        t.lse = p.lse[((int64_t)b * p.H + h) * p.S + qrow];
    };
    QBlock nxt;
    load_blk(wave, nxt);
    settle(nxt.q0); settle(nxt.q1); settle(nxt.do0); settle(nxt.do1); settle(nxt.o0); settle(nxt.o1);
    asm volatile("" : "+v"(nxt.lse));
    for (int qb = wave; qb * 16 < p.S; qb += NW) {
        const int q = qb * 16 + qcol;
        const int qrow = q < p.S ? q : p.S - 1;
        const QBlock cur = nxt;
        if ((qb + NW) * 16 < p.S) load_blk(qb + NW, nxt);
        const bf16x8 qf0 = cur.q0, qf1 = cur.q1, do0 = cur.do0, do1 = cur.do1, o0 = cur.o0, o1 = cur.o1;
        float dl = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)do0[e] * (float)o0[e] + (float)do1[e] * (float)o1[e];
        dl = group_sum(dl);
        const int64_t stat = ((int64_t)b * p.H + h) * p.S + qrow;
        const float nlse = -cur.lse * LOG2E;
        if (g == 0 && q < p.S) p.delta[stat] = dl;

        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // software pipeline over the 32-key steps, as in pass B: S / dP of step u + 1 run under the exp / dS arithmetic of u
        f32x4 sa[2][2], dp[2][2];                      // [parity of u][tile]
        auto s_products = [&](int u, f32x4 (&sa_)[2], f32x4 (&dp_)[2]) {
            bf16x8 kr[2][2], vr[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ds = 0; ds < 2; ++ds) {
                    kr[t][ds] = img_row_frag(kimg, il, 2 * u + t, ds);
                    vr[t][ds] = img_row_frag(vimg, il, 2 * u + t, ds);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, d = f32x4{0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[t][0], qf0, a, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[t][0], do0, d, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kr[t][1], qf1, a, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vr[t][1], do1, d, 0, 0, 0);
                sa_[t] = a; dp_[t] = d;
            }
        };
        s_products(0, sa[0], dp[0]);
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            bf16x8 ktr[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) ktr[dt] = img_tr_frag(kimg, il, u, dt);
            if (u + 1 < NT / 2) s_products(u + 1, sa[(u + 1) & 1], dp[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 ds2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * u + t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(sa[u & 1][t][r] * C2 + nlse);
                    if (CAUSAL || kt >= NT - EDGE) {         // compile-time for the non-causal towers
                        const int key = kt * 16 + g * 4 + r;
                        if (key >= p.S || (CAUSAL && key > q)) pr = 0.f;
                    }
                    ds2[t][r] = pr * (dp[u & 1][t][r] - dl);          // the 1/sqrt(d) factor is applied once, to dq
                }
            }
            const bf16x8 dsf = pack8(ds2[0], ds2[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr[dt], dsf, dq[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        settle(nxt.q0); settle(nxt.q1); settle(nxt.do0); settle(nxt.do1); settle(nxt.o0); settle(nxt.o1);
        asm volatile("" : "+v"(nxt.lse));
        if (q < p.S) {
            bf16_t* dqp = p.dqkv + (row_base + q) * ld + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(dqp + dt * 16) = f32x4_to_bf16x4(dq[dt] * SCALE);
        }
    }
}

// ------------------------------------------------------------------------ backward, pass B: dK, dV
template <int NT, bool CAUSAL, int NW, int EDGE>
__global__ __launch_bounds__(NW * 64, 2) void mha_bwd_dkv_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NT * 16;
    char* qimg = smem;
    char* doimg = smem + SP * 128;
    float* slse = (float*)(smem + 2 * SP * 128);   // -lse * log2e per query
    float* sdel = slse + SP;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;

    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    const bf16_t* dobase = p.dout + row_base * D + h * 64;
    const int64_t remain_o = ((int64_t)(p.batch - b) * p.S * D - h * 64) * 2;
    const uint32_t lim_o = (uint32_t)(remain_o > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain_o);
    dma_image(qimg, uniform_rsrc(base, lim), ld * 2, SP / 8, wave, NW, lane);
    dma_image(doimg, uniform_rsrc(dobase, lim_o), D * 2, SP / 8, wave, NW, lane);
    for (int i = threadIdx.x; i < SP; i += NW * 64) {
        const int64_t stat = ((int64_t)b * p.H + h) * p.S + i;
        slse[i] = i < p.S ? -p.lse[stat] * LOG2E : 0.f;
        sdel[i] = i < p.S ? p.delta[stat] : 0.f;
    }
    __syncthreads();

    const int kcol = lane & 15, g = lane >> 4;
    const ImgLane il = img_lane(lane);
    struct KBlock { bf16x8 k0, k1, v0, v1; };
    auto load_kv = [&](int kb, KBlock& t) {
        const int kk = kb * 16 + kcol;
        const bf16_t* kp = base + (int64_t)(kk < p.S ? kk : p.S - 1) * ld + D + 8 * g;
        t.k0 = *(const bf16x8*)kp;
        t.k1 = *(const bf16x8*)(kp + 32);
        t.v0 = *(const bf16x8*)(kp + D);
        t.v1 = *(const bf16x8*)(kp + D + 32);
    };
    KBlock knxt;
    load_kv(wave, knxt);
    settle(knxt.k0); settle(knxt.k1); settle(knxt.v0); settle(knxt.v1);
    for (int kb = wave; kb * 16 < p.S; kb += NW) {
        const int key = kb * 16 + kcol;
        const KBlock kcur = knxt;
        if ((kb + NW) * 16 < p.S) load_kv(kb + NW, knxt);
        const bf16x8 kf0 = kcur.k0, kf1 = kcur.k1, vf0 = kcur.v0, vf1 = kcur.v1;

        f32x4 dk[4], dv[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // Software pipeline over the 32-query steps: the S / dP products of step u + 1 are issued under the exp / dS
        // arithmetic of step u (the MFMA pipe runs them while the VALU works), and the transposed fragments of step u are
        // requested before that arithmetic, so no step starts with an LDS round trip or waits for its own MFMAs.
        f32x4 sa[2][2], dp[2][2];                      // [parity of u][tile]
        auto s_products = [&](int u, f32x4 (&sa_)[2], f32x4 (&dp_)[2]) {
            bf16x8 qr[2][2], dr[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ds = 0; ds < 2; ++ds) {
                    qr[t][ds] = img_row_frag(qimg, il, 2 * u + t, ds);
                    dr[t][ds] = img_row_frag(doimg, il, 2 * u + t, ds);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f}, d = f32x4{0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[t][0], kf0, a, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dr[t][0], vf0, d, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[t][1], kf1, a, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dr[t][1], vf1, d, 0, 0, 0);
                sa_[t] = a; dp_[t] = d;
            }
        };
        s_products(0, sa[0], dp[0]);
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
            bf16x8 dot[4], qtr[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { dot[dt] = img_tr_frag(doimg, il, u, dt); qtr[dt] = img_tr_frag(qimg, il, u, dt); }
            if (u + 1 < NT / 2) s_products(u + 1, sa[(u + 1) & 1], dp[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 p2[2], ds2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * u + t;
                const f32x4 nl = *(const f32x4*)(slse + qt * 16 + g * 4);
                const f32x4 dl = *(const f32x4*)(sdel + qt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(sa[u & 1][t][r] * C2 + nl[r]);
                    if (CAUSAL || qt >= NT - EDGE) {
                        const int q = qt * 16 + g * 4 + r;
                        if (q >= p.S || (CAUSAL && key > q)) pr = 0.f;
                    }
                    p2[t][r] = pr;
                    ds2[t][r] = pr * (dp[u & 1][t][r] - dl[r]);       // the 1/sqrt(d) factor is applied once, to dk
                }
            }
            const bf16x8 pf = pack8(p2[0], p2[1]);
            const bf16x8 dsf = pack8(ds2[0], ds2[1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[dt], pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], dsf, dk[dt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        settle(knxt.k0); settle(knxt.k1); settle(knxt.v0); settle(knxt.v1);
        if (key < p.S) {
            bf16_t* dkp = p.dqkv + (row_base + key) * ld + D + h * 64 + g * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(bf16x4*)(dkp + dt * 16) = f32x4_to_bf16x4(dk[dt] * SCALE);
                *(bf16x4*)(dkp + D + dt * 16) = f32x4_to_bf16x4(dv[dt]);
            }
        }
    }
}


// ------------------------------------------------------------------------ backward, ONE pass: the scheme (round 3; S <= 320, no mask)
// The two passes above recompute S and dP twice (7 contractions, exp twice).  This kernel does the five contractions once:
//     S^T = Q K^T, dP^T = dO V^T            (key on the MFMA column: this wave's key blocks are the register-resident B operands)
//     dV^T += dO^T P, dK^T += Q^T dS        (the S^T / dP^T accumulators are, after exp and packing, already the B operands)
//     dQ^T += K^T dS^T                      (sums over KEYS, i.e. over the lane index of dS: one transpose, through LDS)
// Work split: ONE workgroup of 4 waves per (batch, head) and CU, one wave per SIMD with the whole 512-register file -- a wave
// keeps dK^T / dV^T of its NT / 4 key blocks (160 accumulator registers at NT = 20) plus their K / V operand fragments for the
// whole problem, so dK and dV need no sum across waves, and the 20 key blocks split evenly over the 4 waves (they do not over 8).
// Per 32-query step u a wave (a) runs S / dP / exp / dS / dV / dK for its key blocks against the step's Q / dO fragments (read
// from the LDS images ONCE per step, not once per key block), (b) leaves its dS tiles in the exchange buffer X[u & 1]
// ([key][32 queries] bf16), and (c) contracts K^T with the WHOLE dS^T of the previous step (X[(u - 1) & 1], all key blocks, read
// back with transposed reads) for its 1/4 of that step's dQ^T tile -- complete sums in a fixed order: no atomics, bitwise
// reproducible.  (c) is pure LDS + MFMA work that the scheduler can slot between the VALU-heavy stages of (a); one workgroup
// barrier per step.  Masking costs nothing: keys >= S start their S accumulator at -1e30 and queries >= S carry -inf in the
// "-lse" row constant, so their P and dS are exactly 0.
// LDS: Q, dO, K images (3 x NT x 2 KiB) + X (2 x NT x 1 KiB) = NT x 8 KiB = 160 KiB at NT = 20.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#ifdef VIPANT_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[64];
#define STAMP(i) do { if (prob == 3000 && lane == 0 && wave == 1) g_attn_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------ backward, ONE pass, streamed operands (round 4)
// The five contractions, regions and issue groups of the scheme above (round 3's kernel with resident Q / dO / K images is kept as
// text: tools/probes/mha_bwd1_resident.hip.txt); what changed in round 4 is where the operands wait.  There a
// problem began with 200 KB of loads (Q, dO, K images, K / V fragments) that nothing overlapped -- one workgroup per CU, 13-16 k of
// 61 k cycles -- because the three images filled the LDS.  Only one 32-query step's rows of Q and dO are ever read at a time, so here
// they pass through a ring of four 8-KiB stages, requested three steps ahead; the LDS that frees holds a SECOND K image, filled one
// piece per step while the current problem runs; V fragments for the next problem are requested when the last A stage has issued;
// delta = rowsum(dO . O) is taken per step from the stage's dO rows and 16 bytes of O per thread and handed over through 128 floats
// of LDS (no global round trip, no 320-row pass up front).  The workgroup is persistent, the ring runs on across problems: a
// problem's first steps, its K image, its statistics are all in place when the previous problem's dK / dV leave.
// LDS: K images 2 x 40 KiB | X 2 x 20 KiB | ring 4 x (Q 4 KiB + dO 4 KiB) | delta 2 x 32 floats | O rows 4 x 1 KiB | lse 4 x 256 B
// (+ 16 B: the ticket) = 157.27 KiB.  NOTHING inside the loop is loaded into registers from global memory: every request is an LDS-DMA piece (inline
// asm, so that hipcc's wait insertion does not see it; explicit counted vmcnt waits).  An asm load into registers whose wait comes
// a step later is not safe -- the register allocator may park the "loaded" value elsewhere before the data has arrived -- and a
// compiler-visible load would be awaited together with every piece in flight.  The O rows and lse words a wave needs for its
// part of delta / its row constants go to buffers private to the wave: no barrier, only the wave's own wait.
// Q8: dK / dV also leave as e4m3 with one scale per 32 columns (p.gq, p.gq_scale: common.h), quantised from the staged rows at the
// problem's end -- 40 more stores per wave behind the 22, so the counted wait there becomes vmcnt(62).
template <int NT, bool Q8 = false>
__global__ __launch_bounds__(256, 1) void mha_bwd1s_kernel(MhaArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SP = NT * 16, KPW = NT / 4, NU = NT / 2;
    static_assert(NT == 20 && KPW == 5, "written for 20 key blocks of 16: five per wave, ten 32-query steps");
    constexpr int KIMG = SP * 128, XB = SP * 64, STG = 8192;
    char* const xbuf = smem + 2 * KIMG;
    char* const ring = xbuf + 2 * XB;
    float* const sdel = (float*)(ring + 4 * STG);      // [2][32]
    char* const obuf = ring + 4 * STG + 256;           // [4 waves][8 rows x 128 B]: the O rows a wave's threads take delta from
    char* const lbuf = obuf + 4096;                    // [4 waves][64 floats]: lse of a step's queries (32 used), one copy per wave
    LDS_AS uint32_t* const tkw = (LDS_AS uint32_t*)(lbuf + 1024);    // the ticket drawn during a problem, for every wave (see the walk below)
    const int lane0 = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int D = p.H * 64, ld = 3 * D;
    const int nprob = p.batch * p.H;

    // ---- what a global step needs from HBM.  Global step = (problem, 32-query step).  Everything per-lane is a constant of the
    // lane (the swizzle of an 8-row block depends on the row's low three bits only) and everything per-step rides the scalar
    // offset of a buffer access: a request costs no vector arithmetic inside the hand-scheduled regions.  A problem's base
    // pointers (one integer division) are made once per problem; past the last problem the descriptors have zero length.
    struct Prob { const bf16_t* q; const bf16_t* dO; const bf16_t* o; const float* lse; uint32_t limq, limdo; uint32_t any; };
    auto make_prob = [&](int pr) {
        Prob t;
        t.any = pr < nprob ? 1u : 0u;
        const int pq = t.any ? pr : 0;
        const int b = pq / p.H, h = pq % p.H;
        t.q = p.qkv + (int64_t)b * p.S * ld + h * 64;
        t.dO = p.dout + (int64_t)b * p.S * D + h * 64;
        t.o = p.out + (int64_t)b * p.S * D + h * 64;
        t.lse = p.lse + (int64_t)pq * p.S;
        const int64_t rq = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2, rd = ((int64_t)(p.batch - b) * p.S * D - h * 64) * 2;
        t.limq = t.any ? (uint32_t)(rq > 0xFFFFFFFFll ? 0xFFFFFFFFll : rq) : 0u;
        t.limdo = t.any ? (uint32_t)(rd > 0xFFFFFFFFll ? 0xFFFFFFFFll : rd) : 0u;
        return t;
    };
    // per-lane constants (recomputed per problem from the laundered lane index)
    struct LaneK { uint32_t vq, vdo, vo, vl; };
    auto make_lanek = [&](int lane) {
        LaneK k;
        const int r8 = lane >> 3, c = (lane & 7) ^ img_swz(r8);
        k.vq = (uint32_t)(r8 * (ld * 2) + c * 16);
        k.vdo = (uint32_t)(r8 * (D * 2) + c * 16);
        k.vo = (uint32_t)(r8 * (D * 2) + (lane & 7) * 16);
        k.vl = (uint32_t)(lane * 4);
        return k;
    };
    // one stage = rows 32 v .. 32 v + 31 of Q (blocks 0-3) and dO (blocks 4-7); wave w brings block w of each: two pieces
    auto stage_pieces = [&](const Prob& t, int v, int slot, const LaneK& k) {
        char* st = ring + slot * STG;
        lds_dma16_asm(uniform_rsrc(t.q, t.limq), st + wave * 1024, k.vq, (uint32_t)((32 * v + 8 * wave) * (ld * 2)));
        lds_dma16_asm(uniform_rsrc(t.dO, t.limdo), st + 4096 + wave * 1024, k.vdo, (uint32_t)((32 * v + 8 * wave) * (D * 2)));
    };
    // K image: 40 blocks of 8 rows, block 4 i + wave is this wave's piece i (0..9)
    auto k_piece = [&](const Prob& t, int i, char* img, const LaneK& k) {
        const int blk = 4 * i + wave;
        lds_dma16_asm(uniform_rsrc(t.q, t.limq), img + blk * 1024, k.vq, (uint32_t)(blk * 8 * (ld * 2) + D * 2));
    };
    // the 8 rows of O this wave's threads take delta from (thread: row 8 wave + (lane >> 3), 8 columns from 8 (lane & 7)), one 1-KiB
    // piece, linear: a thread reads back exactly the 16 bytes its lane brought; rows >= S read as zeros
    auto o_piece = [&](const Prob& t, int v, const LaneK& k) {
        lds_dma16_asm(uniform_rsrc(t.o, t.any ? (uint32_t)(((int64_t)(p.S - 1) * D + 64) * 2) : 0u), obuf + wave * 1024, k.vo,
                      (uint32_t)((32 * v + 8 * wave) * (D * 2)));
    };
    // lse of a step's 32 queries (64 floats fetched, one per lane; rows >= S read as 0 and are masked by the caller)
    auto lse_piece = [&](const Prob& t, int v, const LaneK& k) {
        lds_dma4_asm(uniform_rsrc(t.lse, t.any ? (uint32_t)p.S * 4u : 0u), lbuf + wave * 256, k.vl, (uint32_t)(v * 128));
    };
    // delta of one step from the stage's dO rows and the wave's O rows: 8 lanes share a row; three DPP adds (quad xor 1, quad xor 2,
    // then the mirror of each 8-lane half row, which pairs the two quads)
    auto delta_step = [&](int slot, int dbuf, int tid, int lane) {
        const int row = tid >> 3, c0 = tid & 7;
        const v4i32_t dw = *(const v4i32_t*)(ring + slot * STG + 4096 + row * 128 + ((c0 ^ img_swz(row)) << 4));
        const v4i32_t ow = *(const v4i32_t*)(obuf + wave * 1024 + lane * 16);
        const bf16x8 d0 = __builtin_bit_cast(bf16x8, dw), o0 = __builtin_bit_cast(bf16x8, ow);
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sacc += (float)d0[e] * (float)o0[e];
        // (v_dot2_f32_bf16 in place of the converts and multiply-adds gave wrong sums inside this kernel although a stand-alone probe
        // of the instruction passes -- tools/probes/dot2_dpp_probe.hip -- not pursued; the three DPP adds replace three ds_bpermute
        // round trips, which at one wave per SIMD are exposed latency)
        sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0xB1, 0xF, 0xF, true));
        sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x4E, 0xF, 0xF, true));
        sacc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sacc), 0x141, 0xF, 0xF, true));
        if (c0 == 0) sdel[dbuf * 32 + row] = sacc;
    };

    // ---- the walk over problems.  A problem needs its successor's index at its START (the successor's K image, first stages and
    // V fragments are requested under this problem's arithmetic), so the draw runs one problem ahead of that: problems 0 and 1 of
    // a workgroup are static (blockIdx.x, blockIdx.x + grid); during problem t thread 0 draws the index of problem t + 2 from the
    // stream's counter (one agent-scope atomic, issued before the next problem's V loads and therefore covered by the same counted
    // wait at the problem's end), puts it into LDS before the problem's last barrier, and every wave reads it from there.  A
    // workgroup held up by a co-resident kernel simply draws fewer problems.  Results do not depend on who computes a problem.
    // p.tk == NULL: the static stride.  The atomic is inline asm (result in an accumulation register, as the V loads: the compiler
    // must not see a load it would wait for); all four waves issue it, with EXEC = lane 0 of wave 0 only and EXEC = 0 elsewhere.
    const bool dyn = p.tk != nullptr;
    if (dyn && blockIdx.x == 0 && threadIdx.x < 8) tickets::put(p.tk_other + threadIdx.x, 0u);
    auto draw = [&]() {
        uint32_t raw;
        uint64_t keep;
        const uint32_t who = (uint32_t)__builtin_amdgcn_readfirstlane(dyn && wave == 0 ? 1 : 0);
        asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %3, %4, off sc0\n\ts_mov_b64 exec, %1"
                     : "=&a"(raw), "=&s"(keep) : "s"(who), "v"((uint64_t)p.tk), "a"(1u) : "memory");
        return raw;
    };
    int prob = blockIdx.x;
    int nxt = blockIdx.x + gridDim.x;
    int gs = 0;                                        // global step counter: ring slot = gs & 3, delta buffer = gs & 1
    int cur = 0;                                       // K image in use
    bf16x8 vf[KPW][2];
    auto v_load = [&](const Prob& t, int lane) {
        const int kcol = lane & 15, g = lane >> 4;
#pragma unroll
        for (int j = 0; j < KPW; ++j) {
            const int key = (wave + 4 * j) * 16 + kcol;
            const bf16_t* kp = t.q + (int64_t)(key < p.S ? key : p.S - 1) * ld + 2 * D + 8 * g;      // rows >= S: row S - 1 (finite)
            v4i32_t t0, t1;
            asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64" : "=&a"(t0), "=&a"(t1) : "v"(kp) : "memory");
            vf[j][0] = __builtin_bit_cast(bf16x8, t0); vf[j][1] = __builtin_bit_cast(bf16x8, t1);
        }
    };
    Prob pc = make_prob(prob);
    {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int tid = wave * 64 + lane;
        const LaneK lk = make_lanek(lane);
        for (int i = 0; i < 10; ++i) k_piece(pc, i, smem, lk);
        stage_pieces(pc, 0, 0, lk);
        stage_pieces(pc, 1, 1, lk);
        stage_pieces(pc, 2, 2, lk);
        v_load(pc, lane);
        lse_piece(pc, 0, lk);
        o_piece(pc, 0, lk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        delta_step(0, 0, tid, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the O rows are read: the piece for step 1 may overwrite them
        o_piece(pc, 1, lk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (in the steady state the last step of the previous problem waits for it)
    }

  for (; prob < nprob; cur ^= 1) {
    // (the lane index is laundered per problem: per-lane addresses are recomputed here instead of being kept across the loop)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    const int kcol = lane & 15, g = lane >> 4;
    const LaneK lk = make_lanek(lane);
    const Prob pn = make_prob(nxt);
    char* const kimg = smem + cur * KIMG;
    char* const knext = smem + (cur ^ 1) * KIMG;

    STAMP(0);
    // ---- problem switch: the K image landed during the previous problem (its last piece was awaited at that problem's end)
    for (int i = p.S * 8 + tid; i < SP * 8; i += 256) *(u32x4*)(kimg + i * 16) = u32x4{0u, 0u, 0u, 0u};      // keys >= S
    __syncthreads();                                   // ... and delta of step 0 is visible
    const ImgLane il = img_lane(lane);
    bf16x8 kf[KPW][2];
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        kf[j][0] = img_row_frag(kimg, il, wave + 4 * j, 0);
        kf[j][1] = img_row_frag(kimg, il, wave + 4 * j, 1);
    }
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        asm volatile("" : "+a"(kf[j][0])); asm volatile("" : "+a"(kf[j][1]));
        asm volatile("" : "+a"(vf[j][0])); asm volatile("" : "+a"(vf[j][1]));
    }

    // dQ rows of this (batch, head): stores of rows >= S (and of the step before the first) go out of range and are dropped
    bf16_t* const dq_base = p.dqkv + (pc.q - p.qkv);  // the same (batch, head) offset in the gradient buffer
    const __amdgpu_buffer_rsrc_t rs_dq = uniform_rsrc(dq_base, (uint32_t)(((int64_t)(p.S - 1) * ld + 64) * 2));
    auto load_rows = [&](const char* st, bf16x8 (&qr)[2][2], bf16x8 (&dr)[2][2]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
                qr[tt][ds] = img_row_frag(st, il, tt, ds);
                dr[tt][ds] = img_row_frag(st + 4096, il, tt, ds);
            }
    };
    auto load_tr = [&](const char* st, bf16x8 (&qtr)[4], bf16x8 (&dot)[4]) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { qtr[dt] = img_tr_frag(st, il, 0, dt); dot[dt] = img_tr_frag(st + 4096, il, 0, dt); }
    };
    // row constants of a step: -lse * log2e (queries >= S: -inf) from the raw words, -delta from the hand-over buffer
    auto make_stats = [&](int u, int dbuf, f32x4 (&nl)[2], f32x4 (&ndl)[2]) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int q0 = u * 32 + tt * 16 + g * 4;
            const f32x4 lv = *(const f32x4*)(lbuf + wave * 256 + (tt * 16 + g * 4) * 4);
            const f32x4 dv4 = *(const f32x4*)(sdel + dbuf * 32 + tt * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nl[tt][r] = q0 + r < p.S ? -lv[r] * LOG2E : -INFINITY;
                ndl[tt][r] = -dv4[r];
            }
        }
    };
    const uint32_t xswz = (uint32_t)((kcol >> 2) & 1) * 32u;
    const uint32_t xw0 = (uint32_t)(kcol * 64 + g * 8) + xswz, xw1 = (uint32_t)(kcol * 64 + g * 8) + (32u ^ xswz);
    const int qt2 = wave & 1, dtp = wave >> 1;
    const int xra = 4 * g + ((lane >> 2) & 3);
    const uint32_t xrd = (uint32_t)(xra * 64 + ((qt2 ^ (g & 1)) * 32) + (lane & 3) * 8);
    const uint32_t ka0 = dtp ? il.tr[2] : il.tr[0], ka1 = dtp ? il.tr[3] : il.tr[1];
    auto tr_pair = [&](const char* pa, uint32_t second) {
        const bf16x4 lo = lds_read_tr16(pa);
        const bf16x4 hi = lds_read_tr16(pa + second);
        bf16x8 r;
        r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
        r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
        return r;
    };

    f32x4 dk[KPW][4], dv[KPW][4];
#pragma unroll
    for (int j = 0; j < KPW; ++j)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dk[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    bf16x8 qr[2][2], dr[2][2], qtr[4], dot[4];
    f32x4 nl[2], ndl[2];
    f32x4 sa[2][2], dp[2][2];
    f32x4 p2[2], ds2[2];
    bf16x8 pf_c, dsf_c;
    f32x4 dq0, dq1;
    bf16x8 xfa, k0a, k1a, xfb, k1b, k0b;
    auto s_mfma = [&](auto jc, auto ic) {
        constexpr int j = decltype(jc)::value, i = decltype(ic)::value, tt = i >> 2, par = j & 1;
        if ((i & 3) == 0) sa[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[tt][0], kf[j][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if ((i & 3) == 1) dp[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dr[tt][0], vf[j][0], ndl[tt], 0, 0, 0);
        if ((i & 3) == 2) sa[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qr[tt][1], kf[j][1], sa[par][tt], 0, 0, 0);
        if ((i & 3) == 3) dp[par][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dr[tt][1], vf[j][1], dp[par][tt], 0, 0, 0);
    };
    auto a_mfma = [&](auto jc, auto ic) {
        constexpr int j = decltype(jc)::value, i = decltype(ic)::value, dt = i >> 1;
        if (i & 1) dk[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], dsf_c, dk[j][dt], 0, 0, 0);
        else dv[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[dt], pf_c, dv[j][dt], 0, 0, 0);
    };
    auto p_mfma = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i == 0) dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0a, xfa, dq0, 0, 0, 0);
        if (i == 1) dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1a, xfa, dq1, 0, 0, 0);
        if (i == 2) dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0b, xfb, dq0, 0, 0, 0);
        if (i == 3) dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1b, xfb, dq1, 0, 0, 0);
    };
    auto p_load = [&](auto ic, const char* xr, int uk) {
        constexpr int i = decltype(ic)::value;
        if (i == 0) xfa = tr_pair(xr + uk * 2048, 1024);
        if (i == 1) k0a = tr_pair(kimg + ka0 + uk * 4096, 2048);
        if (i == 2) k1a = tr_pair(kimg + ka1 + uk * 4096, 2048);
        if (i == 3) xfb = tr_pair(xr + (uk + 1) * 2048, 1024);
        if (i == 4) k0b = tr_pair(kimg + ka0 + (uk + 1) * 4096, 2048);
        if (i == 5) k1b = tr_pair(kimg + ka1 + (uk + 1) * 4096, 2048);
    };
    float tf[2];
    auto v_f = [&](auto jc, auto ic) {
        constexpr int j = decltype(jc)::value, i = decltype(ic)::value, tt = i >> 2, r = i & 3, par = j & 1;
        tf[i & 1] = sa[par][tt][r] * C2 + nl[tt][r];
    };
    auto v_e = [&](auto ic) {
        constexpr int i = decltype(ic)::value, tt = i >> 2, r = i & 3;
        p2[tt][r] = __builtin_amdgcn_exp2f(tf[i & 1]);
    };
    auto v_m = [&](auto jc, auto ic) {
        constexpr int j = decltype(jc)::value, i = decltype(ic)::value, tt = i >> 2, r = i & 3, par = j & 1;
        ds2[tt][r] = p2[tt][r] * dp[par][tt][r];
    };
    auto dq_store = [&](int u, bool on) {
        const int q = u * 32 + qt2 * 16 + kcol;
        const uint32_t off = (on && q < p.S) ? (uint32_t)(((int64_t)q * ld + dtp * 32 + g * 4) * 2) : 0xFFFFFFF0u;
        typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f32x4_to_bf16x4(dq0 * SCALE)), rs_dq, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_, f32x4_to_bf16x4(dq1 * SCALE)), rs_dq, off, 32, 0);
    };
    auto region = [&](auto jc, char* xw, const char* xr, const char* st_next) {
        constexpr int J = decltype(jc)::value;
        constexpr int nS = J + 1 < KPW ? 8 : 0, nA = J >= 1 ? 8 : 0;
        bf16x8 pf_n, dsf_n;
        auto mfma_n = [&](auto nc) {
            constexpr int n = decltype(nc)::value;
            if constexpr (n < nS) s_mfma(Int2<(J + 1 < KPW ? J + 1 : 0)>{}, Int2<(n < nS ? n : 0)>{});
            else if constexpr (n < nS + nA) a_mfma(Int2<(J >= 1 ? J - 1 : 0)>{}, Int2<(n - nS) & 7>{});
            else if constexpr (n < nS + nA + 4) p_mfma(Int2<(n - nS - nA) & 3>{});
        };
        auto group = [&](auto gc) {
            constexpr int gi = decltype(gc)::value;
            if constexpr (gi < 6) p_load(Int2<gi>{}, xr, 2 * J);
            mfma_n(Int2<2 * gi>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (gi < 8) v_e(Int2<(gi < 8 ? gi : 0)>{});
            if constexpr (gi >= 1 && gi <= 8) v_m(jc, Int2<(gi >= 1 && gi <= 8 ? gi - 1 : 0)>{});
            if constexpr (gi == 8) pf_n = pack8_pairs(p2[0], p2[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_n(Int2<2 * gi + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (gi < 7) v_f(jc, Int2<(gi < 7 ? gi + 1 : 0)>{});
            if constexpr (gi == 9 && J + 1 < KPW) v_f(Int2<(J + 1 < KPW ? J + 1 : 0)>{}, Int2<0>{});
            if constexpr (gi == 9) dsf_n = pack8_pairs(ds2[0], ds2[1]);
            if constexpr (gi == 9) {
                const u32x4 dw = __builtin_bit_cast(u32x4, dsf_n);
                typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                *(u32x2_t*)(xw + J * 4096 + xw0) = u32x2_t{dw[0], dw[1]};
                *(u32x2_t*)(xw + J * 4096 + xw1) = u32x2_t{dw[2], dw[3]};
                if (J == KPW - 1) load_rows(st_next, qr, dr);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        group(Int2<0>{}); group(Int2<1>{}); group(Int2<2>{}); group(Int2<3>{}); group(Int2<4>{});
        group(Int2<5>{}); group(Int2<6>{}); group(Int2<7>{}); group(Int2<8>{}); group(Int2<9>{});
        pf_c = pf_n; dsf_c = dsf_n;
    };

    // ---- step 0's operands: its stage landed long ago, its statistics came with the previous problem's last step (or the preamble)
    load_rows(ring + (gs & 3) * STG, qr, dr);
    make_stats(0, gs & 1, nl, ndl);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { qtr[dt] = bf16x8{}; dot[dt] = bf16x8{}; }
    pf_c = bf16x8{}; dsf_c = bf16x8{};
    STAMP(1);
    for (int u = 0; u < NU; ++u, ++gs) {
        char* xw = xbuf + (u & 1) * XB + wave * 1024;
        const char* xr = xbuf + ((u & 1) ^ 1) * XB + xrd;              // X of step u - 1
        const char* st = ring + (gs & 3) * STG;
        // which (problem, step) the look-ahead requests of this step belong to
        const Prob& pr1 = u + 1 < NU ? pc : pn; const int v1 = u + 1 < NU ? u + 1 : u + 1 - NU;
        const Prob& pr2 = u + 2 < NU ? pc : pn; const int v2 = u + 2 < NU ? u + 2 : u + 2 - NU;
        const Prob& pr3 = u + 3 < NU ? pc : pn; const int v3 = u + 3 < NU ? u + 3 : u + 3 - NU;
        // head: the previous step's last A stage (zeros at u = 0), this step's transposed fragments, the first S stage
        a_mfma(Int2<KPW - 1>{}, Int2<0>{}); a_mfma(Int2<KPW - 1>{}, Int2<1>{}); a_mfma(Int2<KPW - 1>{}, Int2<2>{}); a_mfma(Int2<KPW - 1>{}, Int2<3>{});
        a_mfma(Int2<KPW - 1>{}, Int2<4>{}); a_mfma(Int2<KPW - 1>{}, Int2<5>{}); a_mfma(Int2<KPW - 1>{}, Int2<6>{}); a_mfma(Int2<KPW - 1>{}, Int2<7>{});
        // delta of the NEXT step from the O rows requested a step ago; then this step's requests, oldest first: lse of the next
        // step, O of the step after next, [this step's three image pieces], [this step's two dQ stores]
        delta_step((gs + 1) & 3, (gs + 1) & 1, tid, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave's O rows and lse words are in registers: their buffers may be refilled
        lse_piece(pr1, v1, lk);
        o_piece(pr2, v2, lk);
        __builtin_amdgcn_sched_barrier(0);
        load_tr(st, qtr, dot);
        s_mfma(Int2<0>{}, Int2<0>{}); s_mfma(Int2<0>{}, Int2<1>{}); s_mfma(Int2<0>{}, Int2<2>{}); s_mfma(Int2<0>{}, Int2<3>{});
        s_mfma(Int2<0>{}, Int2<4>{}); s_mfma(Int2<0>{}, Int2<5>{}); s_mfma(Int2<0>{}, Int2<6>{}); s_mfma(Int2<0>{}, Int2<7>{});
        dq0 = f32x4{0.f, 0.f, 0.f, 0.f}; dq1 = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        v_f(Int2<0>{}, Int2<0>{});
        __builtin_amdgcn_sched_barrier(0);
        const char* st_next = ring + ((gs + 1) & 3) * STG;
        region(Int2<0>{}, xw, xr, st_next);
        // the ring slot of the PREVIOUS step is free (its last readers finished before that step's barrier): rows of global step + 3
        stage_pieces(pr3, v3, (gs + 3) & 3, lk);
        region(Int2<1>{}, xw, xr, st_next);
        k_piece(pn, u, knext, lk);
        region(Int2<2>{}, xw, xr, st_next);
        region(Int2<3>{}, xw, xr, st_next);
        region(Int2<4>{}, xw, xr, st_next);
        dq_store(u - 1, u > 0);
        if (u == 5) STAMP(20);
        // everything but this step's three pieces and two stores: the lse words, the O words, and every older piece
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        make_stats(v1, (gs + 1) & 1, nl, ndl);
        STAMP(2 + u);
    }
    auto a_stage = [&](auto jc, const bf16x8& pf, const bf16x8& dsf) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dv[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot[dt], pf, dv[j][dt], 0, 0, 0);
            dk[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtr[dt], dsf, dk[j][dt], 0, 0, 0);
        }
    };
    auto p_stage = [&](const char* xr, int uk0, int n) {
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const int uk = uk0 + i;
            const bf16x8 xf = tr_pair(xr + uk * 2048, 1024);
            const bf16x8 k0 = tr_pair(kimg + ka0 + uk * 4096, 2048), k1 = tr_pair(kimg + ka1 + uk * 4096, 2048);
            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, xf, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, xf, dq1, 0, 0, 0);
        }
    };
    a_stage(Int2<KPW - 1>{}, pf_c, dsf_c);
    // the V fragments' registers are free: request the next problem's (awaited at the end of this problem, behind the dK / dV stores)
    const uint32_t tk_raw = draw();                    // (older than the V loads: the wait for those covers it)
    v_load(pn, lane);
    dq0 = f32x4{0.f, 0.f, 0.f, 0.f}; dq1 = f32x4{0.f, 0.f, 0.f, 0.f};
    p_stage(xbuf + ((NU - 1) & 1) * XB + xrd, 0, NU);
    dq_store(NU - 1, true);
    STAMP(12);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is done with this K image and with X: they become the staging area
    // dK / dV through LDS so that they leave as whole 128-byte rows, 16 B per lane.  Row (key) = 256 B: dK | dV; 8-byte granules XOR-ed
    // with an even number per key (conflict-free writes, and a 16-byte chunk stays a chunk).  Waves 0, 1: the dead K image; 2, 3: X.
    {
        const __amdgpu_buffer_rsrc_t rs_dkv = uniform_rsrc(dq_base, (uint32_t)(((int64_t)(p.S - 1) * ld + 2 * D + 64) * 2));
        // (Q8) the same (batch, head) offset in the e4m3 buffer, one byte per element; the scale buffer whole (its tiled layout mixes rows)
        const int64_t gq_row0 = (int64_t)(prob / p.H) * p.S;
        const uint32_t gq_col0 = (uint32_t)(prob % p.H) * 64u;
        const __amdgpu_buffer_rsrc_t rs_gq = uniform_rsrc(Q8 ? p.gq + (pc.q - p.qkv) : nullptr, Q8 ? (uint32_t)((int64_t)(p.S - 1) * ld + 2 * D + 64) : 0u);
        const __amdgpu_buffer_rsrc_t rs_gs = uniform_rsrc(Q8 ? p.gq_scale : nullptr,
                                                          Q8 ? (uint32_t)((((int64_t)p.batch * p.S + 127) >> 7) * ((3 * D) >> 7) * 512) : 0u);
        STAMP(13);
        char* stg = (wave < 2 ? kimg : xbuf) + (wave & 1) * (KPW * 4096);
#pragma unroll
        for (int j = 0; j < KPW; ++j)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint32_t sw = (uint32_t)(kcol & 7) << 1;
                *(bf16x4*)(stg + j * 4096 + kcol * 256 + (((uint32_t)(dt * 4 + g) ^ sw) << 3)) = f32x4_to_bf16x4(dk[j][dt] * SCALE);
                *(bf16x4*)(stg + j * 4096 + kcol * 256 + (((uint32_t)(16 + dt * 4 + g) ^ sw) << 3)) = f32x4_to_bf16x4(dv[j][dt]);
            }
#pragma unroll
        for (int j = 0; j < KPW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kr_ = i * 4 + (lane >> 4), c = lane & 15;
                const int key = (wave + 4 * j) * 16 + kr_;
                const bf16x8 v = *(const bf16x8*)(stg + j * 4096 + kr_ * 256 + ((c ^ (kr_ & 7)) << 4));
                // rows >= S: redirected out of the descriptor's range and dropped, so that every wave issues exactly 20 stores
                const uint32_t off = key < p.S ? (uint32_t)(((int64_t)key * ld + D + (c >> 3) * D + (c & 7) * 8) * 2) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_dkv, off, 0, 0);
                if constexpr (Q8) {
                    // the lane's 8 columns are a quarter of a 32-column block: the quad agrees on the exponent; 8 bytes per lane
                    // and one scale byte per quad, both as buffer stores so that every lane issues them (dropped when out of range)
                    const u32x4 w = __builtin_bit_cast(u32x4, v);
                    float sc;
                    const uint32_t sb = mx_scale_byte<4>(mx_absmax2(mx_absmax2(mx_absmax2(mx_absmax2(0u, w[0]), w[1]), w[2]), w[3]), &sc);
                    const uint32_t col = (uint32_t)(D + (c >> 3) * D + (c & 7) * 8);       // inside the problem's rows of [M, 3 D]
                    const uint32_t offq = key < p.S ? (uint32_t)((int64_t)key * ld) + col : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)mx_pack4_bf16(w[0], w[1], sc), (uint32_t)mx_pack4_bf16(w[2], w[3], sc)},
                                                          rs_gq, offq, 0, 0);
                    const uint32_t offs = (key < p.S && (c & 3) == 0)
                        ? (uint32_t)mx_scale_offset(gq_row0 + key, (int)((gq_col0 + col) >> 5), (3 * D) >> 7) : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)sb, rs_gs, offs, 0, 0);
                }
            }
    }
    STAMP(14);
    // the next problem's V fragments (10 loads) went out before this problem's 2 + 20 (Q8: + 40) stores
    if constexpr (Q8) asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); else
    asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
    if (dyn && tid == 0) *tkw = 2u * gridDim.x + tk_raw;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // staging rows read: X may be written again; (the other K image is complete:
    STAMP(15);
    pc = pn;
    prob = nxt;
    nxt = dyn ? __builtin_amdgcn_readfirstlane((int)*tkw) : nxt + (int)gridDim.x;
  }                                                    // its last piece was awaited at the end of step 9)
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming variants for sequences that do not fit the resident scheme (S > 384: e.g. the YAML default stride [16,16] at
// T = 1000 gives S = 428; longer clips go beyond).  Same products, same fragment layouts, but the "other" matrices pass
// through LDS in 64-row chunks (two 16 KiB buffers, the next chunk's LDS-DMA in flight under the current chunk's
// MFMAs) and the forward keeps a running maximum / sum (online softmax).  One workgroup = 64 rows of the "own" dimension
// (4 waves x 16), grid = (batch * heads) x ceil(S / 64); two passes backward as above, no atomics.
constexpr int CH = 64;                         // chunk rows
constexpr int CH_IMG = CH * 128;               // bytes of one 64-row image

__device__ __forceinline__ void dma_chunk(char* img, __amdgpu_buffer_rsrc_t rs, uint32_t ld_bytes, int row0, int wave,
                                          int lane) {
    // 8 blocks of 8 rows, two per wave; the swizzle depends on the row inside the image, the source row is row0 + that
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int blk = wave * 2 + i;
        const int r = blk * 8 + (lane >> 3);
        const int c = (lane & 7) ^ img_swz(r);
        lds_dma16(rs, img + blk * 1024, (uint32_t)(row0 + r) * ld_bytes + (uint32_t)c * 16, 0);
    }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void mha_fwd_stream_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * CH_IMG];   // [buffer][K | V]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (p.S + CH - 1) / CH;
    const int bh = blockIdx.x / nqb, qblk = blockIdx.x % nqb;
    const int b = bh / p.H, h = bh % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;
    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    const auto rsK = uniform_rsrc(base + D, lim > (uint32_t)(D * 2) ? lim - D * 2 : 0);
    const auto rsV = uniform_rsrc(base + 2 * D, lim > (uint32_t)(D * 4) ? lim - D * 4 : 0);

    const int qcol = lane & 15, g = lane >> 4;
    const ImgLane il = img_lane(lane);
    const int q = qblk * CH + wave * 16 + qcol;
    const bf16_t* qp = base + (int64_t)(q < p.S ? q : p.S - 1) * ld + 8 * g;
    const bf16x8 qf0 = *(const bf16x8*)qp, qf1 = *(const bf16x8*)(qp + 32);
    const int nch = CAUSAL ? qblk + 1 : (p.S + CH - 1) / CH;    // causal: keys beyond the block's last query never count

    float m = -INFINITY, l = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    dma_chunk(smem, rsK, ld * 2, 0, wave, lane);
    dma_chunk(smem + CH_IMG, rsV, ld * 2, 0, wave, lane);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();                                   // chunk c has landed; every wave is done with chunk c - 1
        if (c + 1 < nch) {
            char* nb = smem + ((c + 1) & 1) * 2 * CH_IMG;
            dma_chunk(nb, rsK, ld * 2, (c + 1) * CH, wave, lane);
            dma_chunk(nb + CH_IMG, rsV, ld * 2, (c + 1) * CH, wave, lane);
        }
        const char* kimg = smem + (c & 1) * 2 * CH_IMG;
        const char* vimg = kimg + CH_IMG;
        f32x4 sc[4];
        float mc = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(kimg, il, kt, 0), qf0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(kimg, il, kt, 1), qf1, acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = c * CH + kt * 16 + g * 4 + r;
                if (key >= p.S || (CAUSAL && key > q)) acc[r] = -INFINITY;
            }
            mc = fmaxf(mc, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
            sc[kt] = acc;
        }
        mc = group_max(mc);
        const float mn = fmaxf(m, mc);                     // finite from the first chunk on (key 0 is never masked)
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * C2);
        const float mnc = mn * C2;
        float lc = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(sc[kt][r] * C2 - mnc);
                sc[kt][r] = e;
                lc += e;
            }
        l = l * alpha + group_sum(lc);
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = o[dt] * alpha;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 pf = pack8(sc[2 * u], sc[2 * u + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag(vimg, il, u, dt), pf, o[dt], 0, 0, 0);
        }
    }
    if (q < p.S) {
        const float inv = __frcp_rn(l);
        bf16_t* op = p.out + (row_base + q) * D + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(op + dt * 16) = f32x4_to_bf16x4(o[dt] * inv);
        if (g == 0) p.lse[((int64_t)b * p.H + h) * p.S + q] = m * SCALE + __logf(l);
    }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void mha_bwd_dq_stream_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * CH_IMG];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqb = (p.S + CH - 1) / CH;
    const int bh = blockIdx.x / nqb, qblk = blockIdx.x % nqb;
    const int b = bh / p.H, h = bh % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;
    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    const auto rsK = uniform_rsrc(base + D, lim > (uint32_t)(D * 2) ? lim - D * 2 : 0);
    const auto rsV = uniform_rsrc(base + 2 * D, lim > (uint32_t)(D * 4) ? lim - D * 4 : 0);
    dma_chunk(smem, rsK, ld * 2, 0, wave, lane);
    dma_chunk(smem + CH_IMG, rsV, ld * 2, 0, wave, lane);

    const int qcol = lane & 15, g = lane >> 4;
    const ImgLane il = img_lane(lane);
    const int q = qblk * CH + wave * 16 + qcol;
    const int qrow = q < p.S ? q : p.S - 1;
    const bf16_t* qp = base + (int64_t)qrow * ld + 8 * g;
    const bf16x8 qf0 = *(const bf16x8*)qp, qf1 = *(const bf16x8*)(qp + 32);
    const int64_t orow = (row_base + qrow) * D + h * 64 + 8 * g;
    const bf16x8 do0 = *(const bf16x8*)(p.dout + orow), do1 = *(const bf16x8*)(p.dout + orow + 32);
    const bf16x8 o0 = *(const bf16x8*)(p.out + orow), o1 = *(const bf16x8*)(p.out + orow + 32);
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)do0[e] * (float)o0[e] + (float)do1[e] * (float)o1[e];
    dl = group_sum(dl);
    const int64_t stat = ((int64_t)b * p.H + h) * p.S + qrow;
    const float nlse = -p.lse[stat] * LOG2E;
    if (g == 0 && q < p.S) p.delta[stat] = dl;
    const int nch = CAUSAL ? qblk + 1 : (p.S + CH - 1) / CH;

    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; ++c) {
        __syncthreads();
        if (c + 1 < nch) {
            char* nb = smem + ((c + 1) & 1) * 2 * CH_IMG;
            dma_chunk(nb, rsK, ld * 2, (c + 1) * CH, wave, lane);
            dma_chunk(nb + CH_IMG, rsV, ld * 2, (c + 1) * CH, wave, lane);
        }
        const char* kimg = smem + (c & 1) * 2 * CH_IMG;
        const char* vimg = kimg + CH_IMG;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 ds2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * u + t;
                f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(kimg, il, kt, 0), qf0, sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(vimg, il, kt, 0), do0, dp, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(kimg, il, kt, 1), qf1, sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(vimg, il, kt, 1), do1, dp, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = c * CH + kt * 16 + g * 4 + r;
                    float pr = __builtin_amdgcn_exp2f(sa[r] * C2 + nlse);
                    if (key >= p.S || (CAUSAL && key > q)) pr = 0.f;
                    ds2[t][r] = pr * (dp[r] - dl);
                }
            }
            const bf16x8 dsf = pack8(ds2[0], ds2[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag(kimg, il, u, dt), dsf, dq[dt], 0, 0, 0);
        }
    }
    if (q < p.S) {
        bf16_t* dqp = p.dqkv + (row_base + q) * ld + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(dqp + dt * 16) = f32x4_to_bf16x4(dq[dt] * SCALE);
    }
}

template <bool CAUSAL>
__global__ __launch_bounds__(256, 2) void mha_bwd_dkv_stream_kernel(MhaArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * CH_IMG];
    __shared__ float sstat[2][2][CH];                      // [buffer][-lse*log2e | delta][query in chunk]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkb = (p.S + CH - 1) / CH;
    const int bh = blockIdx.x / nkb, kblk = blockIdx.x % nkb;
    const int b = bh / p.H, h = bh % p.H;
    const int D = p.H * 64, ld = 3 * D;
    const int64_t row_base = (int64_t)b * p.S;
    const bf16_t* base = p.qkv + row_base * ld + h * 64;
    const int64_t remain = ((int64_t)(p.batch - b) * p.S * ld - h * 64) * 2;
    const uint32_t lim = (uint32_t)(remain > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain);
    const bf16_t* dobase = p.dout + row_base * D + h * 64;
    const int64_t remain_o = ((int64_t)(p.batch - b) * p.S * D - h * 64) * 2;
    const uint32_t lim_o = (uint32_t)(remain_o > 0xFFFFFFFFll ? 0xFFFFFFFFll : remain_o);
    const auto rsQ = uniform_rsrc(base, lim);
    const auto rsO = uniform_rsrc(dobase, lim_o);
    const float* lse_h = p.lse + ((int64_t)b * p.H + h) * p.S;
    const float* del_h = p.delta + ((int64_t)b * p.H + h) * p.S;
    const int c0 = CAUSAL ? kblk : 0;                      // causal: queries before the block's first key never see it
    const int nch = (p.S + CH - 1) / CH;
    auto load_chunk = [&](int c) {
        char* nb = smem + (c & 1) * 2 * CH_IMG;
        dma_chunk(nb, rsQ, ld * 2, c * CH, wave, lane);
        dma_chunk(nb + CH_IMG, rsO, D * 2, c * CH, wave, lane);
        if (threadIdx.x < CH) {
            const int qq = c * CH + threadIdx.x;
            sstat[c & 1][0][threadIdx.x] = qq < p.S ? -lse_h[qq] * LOG2E : 0.f;
            sstat[c & 1][1][threadIdx.x] = qq < p.S ? del_h[qq] : 0.f;
        }
    };
    load_chunk(c0);

    const int kcol = lane & 15, g = lane >> 4;
    const ImgLane il = img_lane(lane);
    const int key = kblk * CH + wave * 16 + kcol;
    const bf16_t* kp = base + (int64_t)(key < p.S ? key : p.S - 1) * ld + D + 8 * g;
    const bf16x8 kf0 = *(const bf16x8*)kp, kf1 = *(const bf16x8*)(kp + 32);
    const bf16x8 vf0 = *(const bf16x8*)(kp + D), vf1 = *(const bf16x8*)(kp + D + 32);

    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int c = c0; c < nch; ++c) {
        __syncthreads();
        if (c + 1 < nch) load_chunk(c + 1);
        const char* qimg = smem + (c & 1) * 2 * CH_IMG;
        const char* doimg = qimg + CH_IMG;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f32x4 p2[2], ds2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * u + t;
                f32x4 sa = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(qimg, il, qt, 0), kf0, sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(doimg, il, qt, 0), vf0, dp, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(qimg, il, qt, 1), kf1, sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag(doimg, il, qt, 1), vf1, dp, 0, 0, 0);
                const f32x4 nl = *(const f32x4*)(&sstat[c & 1][0][qt * 16 + g * 4]);
                const f32x4 dl = *(const f32x4*)(&sstat[c & 1][1][qt * 16 + g * 4]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = c * CH + qt * 16 + g * 4 + r;
                    float pr = __builtin_amdgcn_exp2f(sa[r] * C2 + nl[r]);
                    if (qq >= p.S || (CAUSAL && key > qq)) pr = 0.f;
                    p2[t][r] = pr;
                    ds2[t][r] = pr * (dp[r] - dl[r]);
                }
            }
            const bf16x8 pf = pack8(p2[0], p2[1]);
            const bf16x8 dsf = pack8(ds2[0], ds2[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag(doimg, il, u, dt), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag(qimg, il, u, dt), dsf, dk[dt], 0, 0, 0);
            }
        }
    }
    if (key < p.S) {
        bf16_t* dkp = p.dqkv + (row_base + key) * ld + D + h * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *(bf16x4*)(dkp + dt * 16) = f32x4_to_bf16x4(dk[dt] * SCALE);
            *(bf16x4*)(dkp + D + dt * 16) = f32x4_to_bf16x4(dv[dt]);
        }
    }
}

template <bool CAUSAL, bool BWD>
int32_t launch_stream(const MhaArgs& a, hipStream_t s) {
    const int64_t grid = (int64_t)a.batch * a.H * ((a.S + CH - 1) / CH);
    VIPANT_REQUIRE(grid < (1ll << 31), VIPANT_EBADSHAPE, "mha: too many workgroups");
    if (!BWD) {
        hipLaunchKernelGGL(mha_fwd_stream_kernel<CAUSAL>, dim3((unsigned)grid), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(mha_bwd_dq_stream_kernel<CAUSAL>, dim3((unsigned)grid), dim3(256), 0, s, a);
        VIPANT_LAUNCH_CHECK();
        hipLaunchKernelGGL(mha_bwd_dkv_stream_kernel<CAUSAL>, dim3((unsigned)grid), dim3(256), 0, s, a);
    }
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int NT>
int32_t launch_bwd1s(const MhaArgs& a, hipStream_t s);

// No environment switch selects a kernel here (round 5): every kernel in this file is the only one for its shapes, and
// tests/test_kernels_gpu.py::test_mha reaches each of them -- forward (S <= 384), single-pass streamed backward (224 < S <= 320, no
// mask), two-pass backward (causal / other S), streaming kernels (S > 384).  Variants that were measured and not kept are text
// under tools/probes/ (mha_bwd1_resident, attention_wide, mha_fwd2_variant).

template <int NT, bool CAUSAL, int NW, int EDGE>
int32_t launch_fwd_nw(const MhaArgs& a, hipStream_t s) {
    constexpr int lds = NT * 16 * 128 * 2;
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_fwd_kernel<NT, CAUSAL, NW, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done_on_device(once);
    }
    hipLaunchKernelGGL((mha_fwd_kernel<NT, CAUSAL, NW, EDGE>), dim3(a.batch * a.H), dim3(NW * 64), lds, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int NT, bool CAUSAL, int NW, int EDGE>
int32_t launch_bwd_nw(const MhaArgs& a, hipStream_t s) {
    constexpr int lds_a = NT * 16 * 128 * 2;
    constexpr int lds_b = NT * 16 * 128 * 2 + NT * 16 * 8;
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd_dq_kernel<NT, CAUSAL, NW, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd_dkv_kernel<NT, CAUSAL, NW, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
        done_on_device(once);
    }
    hipLaunchKernelGGL((mha_bwd_dq_kernel<NT, CAUSAL, NW, EDGE>), dim3(a.batch * a.H), dim3(NW * 64), lds_a, s, a);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL((mha_bwd_dkv_kernel<NT, CAUSAL, NW, EDGE>), dim3(a.batch * a.H), dim3(NW * 64), lds_b, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int NT>
int32_t launch_bwd1s(const MhaArgs& a, hipStream_t s) {
    constexpr int lds = 2 * NT * 16 * 128 + 2 * NT * 16 * 64 + 4 * 8192 + 256 + 4096 + 1024 + 16;
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd1s_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd1s_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done_on_device(once);
    }
    const int nprob = a.batch * a.H, cus = device_cus();       // one persistent workgroup per CU
    MhaArgs t = a;
    // ticket walk when a workgroup has problems beyond its two static ones (bit 22 of VIPANT_GEMM_VARIANT: static, for A/B runs)
    const char* var = getenv("VIPANT_GEMM_VARIANT");
    if (nprob > 2 * cus && !(var && (atoi(var) & 4194304))) {
        t.tk = vipant_ticket_block(s, &t.tk_other);
        if (!t.tk) return VIPANT_EHIP;
    }
    if (t.gq != nullptr) hipLaunchKernelGGL((mha_bwd1s_kernel<NT, true>), dim3(nprob < cus ? nprob : cus), dim3(256), lds, s, t);
    else hipLaunchKernelGGL((mha_bwd1s_kernel<NT, false>), dim3(nprob < cus ? nprob : cus), dim3(256), lds, s, t);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

// EDGE = 2 when only the last two key tiles can be partial (S > (NT - 2) * 16), else every tile carries mask code.
template <int NT, bool CAUSAL>
int32_t launch_fwd(const MhaArgs& a, hipStream_t s) {
    // measured (tools/attn_bench.py, b=512 S=316): forward 4 / 5 / 6 / 8 waves = 330 / 461 / 403 / 360 us (the kernels are
    // LDS-instruction bound, more waves only add contention); backward is faster with 8
    const bool tight = a.S > (NT - 2) * 16;
    return tight ? launch_fwd_nw<NT, CAUSAL, 4, (NT < 2 ? NT : 2)>(a, s) : launch_fwd_nw<NT, CAUSAL, 4, NT>(a, s);
}

template <int NT, bool CAUSAL>
int32_t launch_bwd(const MhaArgs& a, hipStream_t s) {
    const bool tight = a.S > (NT - 2) * 16;
    if constexpr (NT == 20 && !CAUSAL) return launch_bwd1s<NT>(a, s);       // single pass, streamed operands
    if (NT >= 8)
        return tight ? launch_bwd_nw<NT, CAUSAL, 8, 2>(a, s) : launch_bwd_nw<NT, CAUSAL, 8, NT>(a, s);
    return tight ? launch_bwd_nw<NT, CAUSAL, 4, (NT < 2 ? NT : 2)>(a, s) : launch_bwd_nw<NT, CAUSAL, 4, NT>(a, s);
}

template <bool CAUSAL, bool BWD>
int32_t dispatch(const MhaArgs& a, hipStream_t s) {
#define VIPANT_MHA_CASE(NT) \
    if (a.S <= NT * 16) return BWD ? launch_bwd<NT, CAUSAL>(a, s) : launch_fwd<NT, CAUSAL>(a, s);
    VIPANT_MHA_CASE(2) VIPANT_MHA_CASE(4) VIPANT_MHA_CASE(6) VIPANT_MHA_CASE(10) VIPANT_MHA_CASE(14)
    VIPANT_MHA_CASE(20) VIPANT_MHA_CASE(24)
#undef VIPANT_MHA_CASE
    return launch_stream<CAUSAL, BWD>(a, s);       // S > 384: chunks of the other dimension streamed through LDS
}

int32_t check(const void* qkv, int64_t batch, int64_t S, int64_t H) {
    VIPANT_REQUIRE(batch > 0 && S > 0 && H > 0, VIPANT_EBADSHAPE, "mha: empty problem");
    VIPANT_REQUIRE((uintptr_t)qkv % 16 == 0, VIPANT_EALIGN, "mha: qkv must be 16-byte aligned");
    VIPANT_REQUIRE(batch * H < (1ll << 31), VIPANT_EBADSHAPE, "mha: too many (batch, head) problems");
    return VIPANT_OK;
}

}  // namespace

extern "C" int32_t vipant_mha_fwd(const uint16_t* qkv, uint16_t* out, float* lse, int64_t batch, int64_t S, int64_t H,
                                  int32_t causal, void* stream) {
    if (int32_t e = check(qkv, batch, S, H)) return e;
    MhaArgs a{(const bf16_t*)qkv, (bf16_t*)out, lse, nullptr, nullptr, nullptr, (int)batch, (int)S, (int)H, 0};
    return causal ? dispatch<true, false>(a, (hipStream_t)stream) : dispatch<false, false>(a, (hipStream_t)stream);
}

extern "C" int32_t vipant_mha_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse,
                                  float* delta, uint16_t* dqkv, int64_t batch, int64_t S, int64_t H, int32_t causal,
                                  void* stream) {
    if (int32_t e = check(qkv, batch, S, H)) return e;
    MhaArgs a{(const bf16_t*)qkv, (bf16_t*)out, const_cast<float*>(lse), (const bf16_t*)dout, delta, (bf16_t*)dqkv,
              (int)batch, (int)S, (int)H, 0};
    return causal ? dispatch<true, true>(a, (hipStream_t)stream) : dispatch<false, true>(a, (hipStream_t)stream);
}

// The same two with the e4m3 + MX-scale form of the result beside the bf16 one (BASELINE configs[4]: the out_proj / in_proj^T
// contractions read it; csrc/block.hip).  The streamed single-pass backward emits it from its problem tail for the dK | dV columns,
// which leave through LDS as whole rows (dQ leaves inside the hand-scheduled steps and takes the stand-alone pass, a third of the
// bytes); every other backward shape, and the forward, run the stand-alone pass over the whole result.  (The forward's emission was
// built and measured in round 5 -- bit-identical, 744 -> 985 us at the ViT-L shape against 744 + 197 for kernel + pass: the kernel is
// issue-bound and its 16-byte row segments make poor stores -- and taken out again; profiles/r5_cfg5_mx.md.)
extern "C" int32_t vipant_mha_fwd_e4m3(const uint16_t* qkv, uint16_t* out, float* lse, uint8_t* oq, uint8_t* oq_scale, int64_t batch,
                                       int64_t S, int64_t H, int32_t causal, void* stream) {
    if (int32_t e = check(qkv, batch, S, H)) return e;
    VIPANT_REQUIRE(oq != nullptr && oq_scale != nullptr && H % 2 == 0, VIPANT_EBADSHAPE, "mha_fwd_e4m3: need both outputs and an even head count");
    MhaArgs a{(const bf16_t*)qkv, (bf16_t*)out, lse, nullptr, nullptr, nullptr, (int)batch, (int)S, (int)H, 0};
    if (int32_t e = causal ? dispatch<true, false>(a, (hipStream_t)stream) : dispatch<false, false>(a, (hipStream_t)stream)) return e;
    // (block-uniform scales, round 6: the same pass, and the weight-gradient contraction of out_proj reads the form as it is)
    return vipant_quant_e4m3_mx32(out, H * 64, oq, H * 64, oq_scale, batch * S, H * 64, stream);
}

extern "C" int32_t vipant_mha_bwd_e4m3(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse, float* delta,
                                       uint16_t* dqkv, uint8_t* gq, uint8_t* gq_scale, int64_t batch, int64_t S, int64_t H,
                                       int32_t causal, void* stream) {
    if (int32_t e = check(qkv, batch, S, H)) return e;
    VIPANT_REQUIRE(gq != nullptr && gq_scale != nullptr && H % 2 == 0, VIPANT_EBADSHAPE, "mha_bwd_e4m3: need both outputs and an even head count");
    MhaArgs a{(const bf16_t*)qkv, (bf16_t*)out, const_cast<float*>(lse), (const bf16_t*)dout, delta, (bf16_t*)dqkv,
              (int)batch, (int)S, (int)H, 0};
    const int64_t D = H * 64;
    const bool fused = !causal && S > 224 && S <= 320;         // mha_bwd1s_kernel (launch_bwd)
    if (fused) { a.gq = gq; a.gq_scale = gq_scale; }
    if (int32_t e = causal ? dispatch<true, true>(a, (hipStream_t)stream) : dispatch<false, true>(a, (hipStream_t)stream)) return e;
    // (the columns this pass makes -- dQ, or all three thirds -- get block-uniform scales; the streamed kernel's dK | dV stay row-wise)
    return vipant_quant_e4m3_mx32_cols(dqkv, 3 * D, gq, 3 * D, gq_scale, batch * S, fused ? D : 3 * D, 3 * D / 128, 0, stream);
}

#ifdef VIPANT_ATTN_STAMPS
extern "C" int32_t vipant_debug_attn_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
