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

#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCALE = 0.125f;              // 1/sqrt(64)
constexpr float C2 = SCALE * LOG2E;

__device__ __forceinline__ int img_swz(int r) { return ((r >> 1) & 3) << 1; }

// Fill a [rows8*8 x 64] bf16 LDS image from `rows8*8` consecutive rows (stride ld_bytes) of a buffer.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, uint32_t bytes) {
    // descriptor inputs made provably wave-uniform, otherwise hipcc wraps every buffer op in a waterfall loop
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return make_rsrc((const void*)(((uint64_t)hi << 32) | lo), (uint32_t)__builtin_amdgcn_readfirstlane(bytes));
}

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

struct MhaArgs {
    const bf16_t* qkv; bf16_t* out; float* lse;
    const bf16_t* dout; float* delta; bf16_t* dqkv;
    int batch, S, H;
};

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
                m = fmaxf(m, fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
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

int attn_waves() {
    static const int nw = getenv("VIPANT_ATTN_WAVES") ? atoi(getenv("VIPANT_ATTN_WAVES")) : 8;
    return nw == 4 ? 4 : 8;
}

template <int NT, bool CAUSAL, int NW, int EDGE>
int32_t launch_fwd_nw(const MhaArgs& a, hipStream_t s) {
    constexpr int lds = NT * 16 * 128 * 2;
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_fwd_kernel<NT, CAUSAL, NW, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    hipLaunchKernelGGL((mha_fwd_kernel<NT, CAUSAL, NW, EDGE>), dim3(a.batch * a.H), dim3(NW * 64), lds, s, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int NT, bool CAUSAL, int NW, int EDGE>
int32_t launch_bwd_nw(const MhaArgs& a, hipStream_t s) {
    constexpr int lds_a = NT * 16 * 128 * 2;
    constexpr int lds_b = NT * 16 * 128 * 2 + NT * 16 * 8;
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd_dq_kernel<NT, CAUSAL, NW, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)mha_bwd_dkv_kernel<NT, CAUSAL, NW, EDGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
        configured = true;
    }
    hipLaunchKernelGGL((mha_bwd_dq_kernel<NT, CAUSAL, NW, EDGE>), dim3(a.batch * a.H), dim3(NW * 64), lds_a, s, a);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL((mha_bwd_dkv_kernel<NT, CAUSAL, NW, EDGE>), dim3(a.batch * a.H), dim3(NW * 64), lds_b, s, a);
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
    if (NT >= 8 && attn_waves() == 8)
        return tight ? launch_bwd_nw<NT, CAUSAL, 8, 2>(a, s) : launch_bwd_nw<NT, CAUSAL, 8, NT>(a, s);
    return tight ? launch_bwd_nw<NT, CAUSAL, 4, (NT < 2 ? NT : 2)>(a, s) : launch_bwd_nw<NT, CAUSAL, 4, NT>(a, s);
}

template <bool CAUSAL, bool BWD>
int32_t dispatch(const MhaArgs& a, hipStream_t s) {
    static const bool force_stream = getenv("VIPANT_ATTN_STREAM") && atoi(getenv("VIPANT_ATTN_STREAM")) == 1;   // tests / timing
    if (force_stream) return launch_stream<CAUSAL, BWD>(a, s);
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
    MhaArgs a{(const bf16_t*)qkv, (bf16_t*)out, lse, nullptr, nullptr, nullptr, (int)batch, (int)S, (int)H};
    return causal ? dispatch<true, false>(a, (hipStream_t)stream) : dispatch<false, false>(a, (hipStream_t)stream);
}

extern "C" int32_t vipant_mha_bwd(const uint16_t* qkv, const uint16_t* out, const uint16_t* dout, const float* lse,
                                  float* delta, uint16_t* dqkv, int64_t batch, int64_t S, int64_t H, int32_t causal,
                                  void* stream) {
    if (int32_t e = check(qkv, batch, S, H)) return e;
    MhaArgs a{(const bf16_t*)qkv, (bf16_t*)out, const_cast<float*>(lse), (const bf16_t*)dout, delta, (bf16_t*)dqkv,
              (int)batch, (int)S, (int)H};
    return causal ? dispatch<true, true>(a, (hipStream_t)stream) : dispatch<false, true>(a, (hipStream_t)stream);
}
