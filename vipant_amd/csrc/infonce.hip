// Symmetric InfoNCE (CELossHead.forward, cvap/module/decoder/loss_head.py:265-284):
//   s = min(exp(logit_scale), scale_max);  Z = s * x1 x2^T;  loss = mean_i CE(Z[i,:], i) + mean_i CE(Z[:,i], i).
//
// gfx950 design: the B x B logits are never written to HBM.
//   pass 1  256x256 logit tiles on MFMA (the shared NT main loop); the epilogue reduces each tile to per-row
//           and per-column (max, sum exp, sum exp*z) triples with wave shuffles and stores only those
//           (B * #tiles floats instead of B^2) plus the diagonal;
//   final   two small kernels merge the triples into row / column log-sum-exp, the loss and d logit_scale
//           ( = sum dZ * Z, expressible from the triples );
//   pass 2  recomputes the tiles that touch this rank's column or row strip, forms s * dZ = s * (softmax_row + softmax_col - 2 I) / B
//           in registers once and stores it as bf16 in both orientations: from the accumulator layout, and TRANSPOSED through an
//           LDS image of the tile (round 2 launched the kernel twice, the second time with the operands swapped);
//   grads   dx2 = (s dZ)^T x1 and dx1 = (s dZ^T)^T x2 are then the same token-reduction (TN) contraction, split over all
//           256 CUs (the row-major form of dx1, an NT contraction with 32 output tiles, left 7/8 of the chip idle).
// Precision: inputs are fp32; pass 1 (the loss, the LSE vectors, d logit_scale) uses a hi/lo bf16 split of both operands
// (x = hi + lo, three MFMA terms hi*hi + hi*lo + lo*hi concatenated along K) with fp32 accumulation: logits to ~1e-6, the
// loss well inside the 1e-3 budget.  Pass 2 needs that only where softmax - I cancels, i.e. on the diagonal ELEMENTS, which reuse
// pass 1's logits; everywhere else one bf16 term (K = E) moves a logit by ~2e-3 * s / 14, below the bf16 rounding of dZ itself.
// exp / log in fp32.
#include "nt_core.h"
#include <stdlib.h>

namespace {

using namespace ntcore;

struct NceWs {       // carved out of the caller's workspace; all offsets 256-B aligned
    bf16_t *x1cat, *x2cat, *dz, *dzt;
    float *rmax, *rsum, *rwz, *cmax, *csum, *cwz, *diag, *rlse, *clse, *scal, *part, *dxtmp, *dxtmp2;   // scal[0] = s, scal[1] = clamped
    void* tn_ws;
    size_t tn_bytes, total;
    int Bp, rparts, cparts;
    // the row-block path (B <= ROWS_MAX_B): transposed bf16 copies of x1 / x2 [E][Bp32], the fp32 logits of both sides
    // [2][B][Bp128], per key-chunk partial (max, sum, sum * z) [2][KC][B] x 3, loss / dlogit_scale accumulators + ticket
    bf16_t *x1t, *x2t;
    float *zws, *pmax, *psum, *pwz, *accum;
    int Bp32, Bp128, KC;
    // the strip path (B > ROWS_MAX_B, gradients for a strip of <= ROWS_MAX_B rows: what one rank of an N-GPU step asks for): pass 1
    // leaves the fp32-grade logits of the strip's rows of Z (zrow [nrows][Bp128]) and of Z^T (zcol [nrows][Bp128]) as it passes them
    float *zrow, *zcol;
    int zrow0, znrows;
};
constexpr int ROWS_MAX_B = 768, ROWS_E = 512;      // 54 us against 114 at B = 512; at B = 1024 the tile kernels win (117 against 135)

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// strip_rows > 0: the layout of the strip path (no B x B gradient matrices, no split-reduction partials)
NceWs carve(char* base, int64_t B, int64_t E, int64_t strip_rows = 0) {
    NceWs w;
    const int64_t Bp = (B + 63) / 64 * 64;
    const int ntm = (int)ceil_div(B, BM), ntn = (int)ceil_div(B, BN);
    w.Bp = (int)Bp; w.rparts = ntn * 4; w.cparts = ntm * 2;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += align256(bytes); return p; };
    w.x1cat = (bf16_t*)take((size_t)B * 3 * E * 2);
    w.x2cat = (bf16_t*)take((size_t)B * 3 * E * 2);
    w.dz = (bf16_t*)take(strip_rows ? 0 : (size_t)B * Bp * 2);
    w.dzt = (bf16_t*)take(strip_rows ? 0 : (size_t)B * Bp * 2);
    w.rmax = (float*)take((size_t)w.rparts * B * 4); w.rsum = (float*)take((size_t)w.rparts * B * 4);
    w.rwz = (float*)take((size_t)w.rparts * B * 4);
    w.cmax = (float*)take((size_t)w.cparts * B * 4); w.csum = (float*)take((size_t)w.cparts * B * 4);
    w.cwz = (float*)take((size_t)w.cparts * B * 4);
    w.diag = (float*)take((size_t)B * 4); w.rlse = (float*)take((size_t)B * 4); w.clse = (float*)take((size_t)B * 4);
    w.scal = (float*)take(256);
    w.part = (float*)take((size_t)ceil_div(B, 16) * 2 * 4);
    w.dxtmp = (float*)take(strip_rows ? 0 : (size_t)(B + 8) * E * 4);          // gradient rows of a strip that does not start on a multiple of 8
    w.dxtmp2 = (float*)take(strip_rows ? 0 : (size_t)(B + 8) * E * 4);
    w.tn_bytes = 0;
    if (!strip_rows) {
        w.tn_bytes = vipant_gemm_tn_workspace_bytes(B, B, E);
        const size_t pair_bytes = vipant_gemm_tn_pair_workspace_bytes(B, B, E);
        if (pair_bytes > w.tn_bytes) w.tn_bytes = pair_bytes;
    }
    w.tn_ws = take(w.tn_bytes);
    w.Bp32 = (int)((B + 31) / 32 * 32); w.Bp128 = (int)((B + 127) / 128 * 128); w.KC = w.Bp128 / 128;
    const bool rows = (B <= ROWS_MAX_B || strip_rows) && E == ROWS_E;
    w.x1t = (bf16_t*)take(rows ? (size_t)E * w.Bp32 * 2 : 0);
    w.x2t = (bf16_t*)take(rows ? (size_t)E * w.Bp32 * 2 : 0);
    w.zrow = (float*)take((size_t)strip_rows * w.Bp128 * 4);
    w.zcol = (float*)take((size_t)strip_rows * w.Bp128 * 4);
    if (!strip_rows) w.zrow = w.zcol = nullptr;
    w.zrow0 = 0; w.znrows = 0;
    const bool small = B <= ROWS_MAX_B && E == ROWS_E;
    w.zws = (float*)take(small ? (size_t)2 * B * w.Bp128 * 4 : 0);
    w.pmax = (float*)take(small ? (size_t)2 * w.KC * B * 4 : 0);
    w.psum = (float*)take(small ? (size_t)2 * w.KC * B * 4 : 0);
    w.pwz = (float*)take(small ? (size_t)2 * w.KC * B * 4 : 0);
    w.accum = (float*)take(256);
    w.total = off;
    return w;
}

// x -> [hi | hi | lo] (x1) or [hi | lo | hi] (x2).
__global__ __launch_bounds__(256) void nce_prep_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                       const float* __restrict__ logit_scale, float scale_max, NceWs w,
                                                       int B, int E, int transposed) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float raw = __expf(logit_scale[0]);
        const bool clamped = scale_max > 0.f && raw > scale_max;
        w.scal[0] = clamped ? scale_max : raw;
        w.scal[1] = clamped ? 1.f : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) w.accum[threadIdx.x] = 0.f;       // (row-block path) sums and the ticket
    if (transposed) {       // bf16(x)^T, [E][Bp32], columns B .. Bp32 zero: the gradient products read keys along rows
        const int64_t tt = (int64_t)E * w.Bp32;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tt; i += (int64_t)gridDim.x * 256) {
            const int n = (int)(i % w.Bp32);
            const int64_t e = i / w.Bp32;
            w.x1t[i] = n < B ? (bf16_t)x1[(int64_t)n * E + e] : (bf16_t)0.f;
            w.x2t[i] = n < B ? (bf16_t)x2[(int64_t)n * E + e] : (bf16_t)0.f;
        }
    }
    const int64_t total = (int64_t)B * E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / E;
        const int c = (int)(i % E);
        const float a = x1[i], t = x2[i];
        const bf16_t ah = (bf16_t)a, th = (bf16_t)t;
        const bf16_t al = (bf16_t)(a - (float)ah), tl = (bf16_t)(t - (float)th);
        bf16_t* p1 = w.x1cat + r * 3 * E + c;
        bf16_t* p2 = w.x2cat + r * 3 * E + c;
        p1[0] = ah; p1[E] = ah; p1[2 * E] = al;
        p2[0] = th; p2[E] = tl; p2[2 * E] = th;
    }
}

// bf16(x)^T for both operands in one launch (strip path, B % 32 == 0): 64 x 64 tiles through LDS, coalesced both ways
__global__ __launch_bounds__(256) void nce_transpose_kernel(const float* __restrict__ x1, const float* __restrict__ x2, NceWs w, int B, int E) {
    __shared__ bf16_t tile[64][66];
    const float* src = blockIdx.z == 0 ? x1 : x2;
    bf16_t* dst = blockIdx.z == 0 ? w.x1t : w.x2t;
    const int tr = blockIdx.y * 64, tc = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int gr = tr + r, gc = tc + tx;
        tile[r][tx] = (gr < B && gc < E) ? (bf16_t)src[(int64_t)gr * E + gc] : (bf16_t)0.0f;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int gc = tc + c, gr = tr + tx;
        if (gc < E && gr < w.Bp32) dst[(int64_t)gc * w.Bp32 + gr] = tile[tx][c];
    }
}

template <int PASS>
__global__ __launch_bounds__(512, 2) void nce_tile_kernel(NceWs w, int B, int K, int row0, int nrows, float gscale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;
    const int ntm = (B + BM - 1) / BM, ntn = (B + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = tile / ntn, tn = tile % ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    // pass 2: a tile feeds dx2 (through s dZ, rows m x columns n) when its COLUMNS meet this rank's strip, and dx1 (through the
    // transpose, rows n x columns m) when its ROWS do
    bool cols_hit = true, rows_hit = true;
    if (PASS == 2) {
        cols_hit = w.dz != nullptr && n0 < row0 + nrows && n0 + BN > row0;
        rows_hit = w.dzt != nullptr && m0 < row0 + nrows && m0 + BM > row0;
        if (!cols_hit && !rows_hit) return;
    }
    f32x4 acc[8][4];
    // K = 3E: [hi|hi|lo] . [hi|lo|hi]; pass 2: the first E columns only (hi . hi) -- its diagonal ELEMENTS, the only place where
    // softmax - I cancels, take the fp32-grade logit pass 1 left in w.diag
    const int Kt = PASS == 2 ? K / 3 : K;
    mainloop(smem, w.x1cat, K, B, w.x2cat, K, B, Kt, m0, n0, wave, lane, acc);
    const float s = w.scal[0];
    const int mb = m0 + wm * 128 + frow, nb = n0 + wn * 64 + fq * 4;

    if (PASS == 1) {
        // scale, mask, diagonal
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mb + i * 16, n = nb + j * 16 + r;
                    const float z = (m < B && n < B) ? s * acc[i][j][r] : -INFINITY;
                    acc[i][j][r] = z;
                    if (m == n && m < B) w.diag[m] = z;
                }
        if (w.zrow != nullptr) {        // strip path: the logits of the strip's rows of Z, and of its rows of Z^T, stay in HBM for the gradient kernel
            const int r0 = w.zrow0, r1 = w.zrow0 + w.znrows;
            if (m0 < r1 && m0 + BM > r0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = mb + i * 16;
                    if (m >= r0 && m < r1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (nb + j * 16 < w.Bp128) *(f32x4*)(w.zrow + (int64_t)(m - r0) * w.Bp128 + nb + j * 16) = acc[i][j];
                    }
                }
            }
            if (n0 < r1 && n0 + BN > r0) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = nb + j * 16 + r;
                        if (n >= r0 && n < r1) {
#pragma unroll
                            for (int i = 0; i < 8; ++i)
                                if (mb + i * 16 < w.Bp128) w.zcol[(int64_t)(n - r0) * w.Bp128 + mb + i * 16] = acc[i][j][r];
                        }
                    }
            }
        }
        // rows: over this wave's 64 columns
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float ms = mx == -INFINITY ? 0.f : mx;
            float se = 0.f, sw = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = acc[i][j][r];
                    const float e = __expf(z - ms);
                    se += e;
                    sw += z == -INFINITY ? 0.f : e * z;
                }
            se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
            sw += __shfl_xor(sw, 16, 64); sw += __shfl_xor(sw, 32, 64);
            const int m = mb + i * 16;
            if (fq == 0 && m < B) {
                const int64_t o = (int64_t)(tn * 4 + wn) * B + m;
                w.rmax[o] = mx; w.rsum[o] = se; w.rwz[o] = sw;
            }
        }
        // columns: over this wave's 128 rows
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, acc[i][j][r]);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                const float ms = mx == -INFINITY ? 0.f : mx;
                float se = 0.f, sw = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = acc[i][j][r];
                    const float e = __expf(z - ms);
                    se += e;
                    sw += z == -INFINITY ? 0.f : e * z;
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { se += __shfl_xor(se, o, 64); sw += __shfl_xor(sw, o, 64); }
                const int n = nb + j * 16 + r;
                if (frow == 0 && n < B) {
                    const int64_t o = (int64_t)(tm * 2 + wm) * B + n;
                    w.cmax[o] = mx; w.csum[o] = se; w.cwz[o] = sw;
                }
            }
    } else {
        // s dZ of the tile, formed once; it leaves twice: straight from the accumulator layout (4 consecutive columns per lane)
        // into dz, and -- through an LDS image [n][m] of the tile, 128 KiB, the operand stages are free by now -- as whole
        // 512-byte rows of the TRANSPOSE into dzt.  16-byte chunks of an image row are XOR-swizzled with the row's quad index so
        // that the four column quads a wave writes at once do not share banks.
        const float k = gscale / (float)B;
        __syncthreads();                       // every wave is done reading the operand stages
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n4 = nb + j * 16;
            f32x4 cl;
#pragma unroll
            for (int r = 0; r < 4; ++r) cl[r] = n4 + r < B ? w.clse[n4 + r] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mb + i * 16;
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < B && n4 < w.Bp) {
                    const float rl = w.rlse[m];
                    const float dg = tm == tn ? w.diag[m] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n4 + r;
                        const float z = m == n ? dg : s * acc[i][j][r];
                        const float g = (__expf(z - rl) + __expf(z - cl[r]) - (m == n ? 2.f : 0.f)) * k;
                        d[r] = n < B ? g * s : 0.f;
                    }
                    if (cols_hit) *(bf16x4*)(w.dz + (int64_t)m * w.Bp + n4) = f32x4_to_bf16x4(d);
                }
                if (rows_hit) {
                    const int ml = wm * 128 + i * 16 + frow;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nl = wn * 64 + j * 16 + fq * 4 + r;
                        *(bf16_t*)(smem + nl * 512 + ((((ml >> 3) ^ (((nl >> 2) & 3) << 1)) << 4) | ((ml & 7) << 1))) = (bf16_t)d[r];
                    }
                }
            }
        }
        if (rows_hit) {
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int idx = t * 512 + (int)threadIdx.x;
                const int nl = idx >> 5, ch = idx & 31;
                const int n = n0 + nl, m8 = m0 + ch * 8;
                if (n < B && m8 < w.Bp)
                    *(u32x4*)(w.dzt + (int64_t)n * w.Bp + m8) = *(const u32x4*)(smem + nl * 512 + ((ch ^ (((nl >> 2) & 3) << 1)) << 4));
            }
        }
    }
}

// Merge the per-tile triples.  A workgroup owns 16 row/column indices; thread t = (slice t >> 4, index t & 15) folds every
// 16th partial of its index for both sides, the 16 slices of an index are combined through LDS in a fixed order
// (reproducible), and the workgroup leaves its two partial sums for the one-workgroup kernel below.
// loss = mean(rlse - d) + mean(clse - d);
// dlogit_scale = gscale/B * (sum_i E_row[i] + sum_j E_col[j] - 2 sum_i d_i), zero when the scale is clamped.
__global__ __launch_bounds__(256) void nce_merge_kernel(NceWs w, int B) {
    __shared__ float part[2][3][16][16];      // [side][max, sum, sum*z][slice][index]
    __shared__ float red[2][16];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float* pm = side == 0 ? w.rmax : w.cmax;
        const float* ps = side == 0 ? w.rsum : w.csum;
        const float* pw = side == 0 ? w.rwz : w.cwz;
        const int parts = side == 0 ? w.rparts : w.cparts;
        float mx = -INFINITY, se = 0.f, sw = 0.f;
        if (i < B) {
            for (int p = sl; p < parts; p += 16) {
                const float pmx = pm[(int64_t)p * B + i];
                if (pmx == -INFINITY) continue;
                const float nm = fmaxf(mx, pmx);
                const float fo = mx == -INFINITY ? 0.f : __expf(mx - nm), fn = __expf(pmx - nm);
                se = se * fo + ps[(int64_t)p * B + i] * fn;
                sw = sw * fo + pw[(int64_t)p * B + i] * fn;
                mx = nm;
            }
        }
        part[side][0][sl][il] = mx; part[side][1][sl][il] = se; part[side][2][sl][il] = sw;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float lsum = 0.f, esum = 0.f;
        if (i < B) {
            float lse2[2], ez2[2];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                float mx = -INFINITY;
                for (int q = 0; q < 16; ++q) mx = fmaxf(mx, part[side][0][q][il]);
                float se = 0.f, sw = 0.f;
                for (int q = 0; q < 16; ++q) {
                    const float pmx = part[side][0][q][il];
                    if (pmx == -INFINITY) continue;
                    const float f = __expf(pmx - mx);
                    se += part[side][1][q][il] * f;
                    sw += part[side][2][q][il] * f;
                }
                const float lse = mx + __logf(se);
                (side == 0 ? w.rlse : w.clse)[i] = lse;
                lse2[side] = lse; ez2[side] = sw / se;
            }
            const float d = w.diag[i];
            lsum = (lse2[0] - d) + (lse2[1] - d);
            esum = ez2[0] + ez2[1] - 2.f * d;
        }
        red[0][il] = lsum; red[1][il] = esum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < 16; ++q) { a += red[0][q]; b += red[1][q]; }
        w.part[2 * blockIdx.x] = a; w.part[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(256) void nce_final_kernel(NceWs w, int B, int nparts, float gscale, float* loss, float* dls) {
    __shared__ float red[2][256];
    float lsum = 0.f, esum = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) { lsum += w.part[2 * i]; esum += w.part[2 * i + 1]; }
    red[0][threadIdx.x] = lsum; red[1][threadIdx.x] = esum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss[0] = red[0][0] / (float)B;
        if (dls != nullptr) dls[0] = w.scal[1] != 0.f ? 0.f : red[1][0] * gscale / (float)B;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Row-block path for few clips (B <= 768: the step's own batch on one GPU).  Two launches behind the prep:
//   lse    workgroup (key chunk of 128, block of 16 queries, side): the fp32-grade logits of 16 rows of Z (side 0: queries x1, keys
//          x2) or of Z^T (side 1) against 128 keys -- [hi|hi|lo] . [hi|lo|hi] over K = 3 E as in the tile kernels, fragments straight
//          from global memory (both operands K-contiguous), a wave per 32 keys -- left in HBM (2 B^2 floats: 2 MiB at B = 512) with
//          the chunk's (max, sum exp, sum exp * z) per query;
//   grad   workgroup (block of 16 queries, side): merges the chunks' triples into the two log-sum-exp vectors it needs (the first
//          workgroup of each side also sums the loss and d logit_scale terms; the second to arrive writes them out), forms
//          s dZ = s (softmax_row + softmax_col - 2 I) / B for its 16 rows in registers, 32 keys at a time, as the B operand of
//          dx^T[columns x 16 rows] += x_other^T[columns x keys] . dZ^T -- the other operand from the transposed bf16 copy the prep
//          leaves, so it is K-contiguous too: no LDS staging, no B x B matrix of dZ anywhere.
template <int KS>
__global__ __launch_bounds__(256) void nce_rows_lse_kernel(NceWs w, int B) {
    __shared__ float red[4][3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int kc = blockIdx.x, blk = blockIdx.y, side = blockIdx.z;
    const int K = KS * 32;
    const bf16_t* Q = side == 0 ? w.x1cat : w.x2cat;
    const bf16_t* Kx = side == 0 ? w.x2cat : w.x1cat;
    const int m = 16 * blk + r;
    const bf16_t* qp = Q + (int64_t)(m < B ? m : B - 1) * K + 8 * g;
    const bf16_t* kp[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 128 * kc + 32 * wave + 16 * t + r;
        kp[t] = Kx + (int64_t)(n < B ? n : B - 1) * K + 8 * g;
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    constexpr int HALF = KS / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        bf16x8 qf[HALF];
#pragma unroll
        for (int s = 0; s < HALF; ++s) qf[s] = *(const bf16x8*)(qp + 32 * (half * HALF + s));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 kf[HALF];
#pragma unroll
            for (int s = 0; s < HALF; ++s) kf[s] = *(const bf16x8*)(kp[t] + 32 * (half * HALF + s));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < HALF; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[s], acc[t], 0, 0, 0);
        }
    }
    // lane: query m = 16 blk + r, keys n0 + i of tile t, n0 = 128 kc + 32 wave + 16 t + 4 g
    const float s = w.scal[0];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n0 = 128 * kc + 32 * wave + 16 * t + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + i;
            const float z = (m < B && n < B) ? s * acc[t][i] : -INFINITY;
            acc[t][i] = z;
            mx = fmaxf(mx, z);
            if (side == 0 && m == n && m < B) w.diag[m] = z;
        }
        if (m < B) *(f32x4*)(w.zws + ((int64_t)side * B + m) * w.Bp128 + n0) = acc[t];
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float ms = mx == -INFINITY ? 0.f : mx;
    float se = 0.f, sw = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float z = acc[t][i];
            const float e = __expf(z - ms);
            se += e;
            sw += z == -INFINITY ? 0.f : e * z;
        }
    se += __shfl_xor(se, 16, 64); se += __shfl_xor(se, 32, 64);
    sw += __shfl_xor(sw, 16, 64); sw += __shfl_xor(sw, 32, 64);
    if (g == 0) { red[wave][0][r] = mx; red[wave][1][r] = se; red[wave][2][r] = sw; }
    __syncthreads();
    if (threadIdx.x < 16 && m < B) {        // the four waves' triples of query r, in a fixed order
        float fm = -INFINITY;
        for (int q = 0; q < 4; ++q) fm = fmaxf(fm, red[q][0][r]);
        float fe = 0.f, fw = 0.f;
        for (int q = 0; q < 4; ++q) {
            const float pm = red[q][0][r];
            if (pm == -INFINITY) continue;
            const float f = __expf(pm - fm);
            fe += red[q][1][r] * f;
            fw += red[q][2][r] * f;
        }
        const int64_t o = ((int64_t)side * w.KC + kc) * B + m;
        w.pmax[o] = fm; w.psum[o] = fe; w.pwz[o] = fw;
    }
}

// the chunks' triples of index i on side sd -> (log-sum-exp, sum exp * z / sum exp); fixed order
__device__ __forceinline__ void nce_rows_merge(const NceWs& w, int B, int sd, int i, float* lse, float* ez) {
    float fm = -INFINITY;
    for (int c = 0; c < w.KC; ++c) fm = fmaxf(fm, w.pmax[((int64_t)sd * w.KC + c) * B + i]);
    float fe = 0.f, fw = 0.f;
    for (int c = 0; c < w.KC; ++c) {
        const int64_t o = ((int64_t)sd * w.KC + c) * B + i;
        const float pm = w.pmax[o];
        if (pm == -INFINITY) continue;
        const float f = __expf(pm - fm);
        fe += w.psum[o] * f;
        fw += w.pwz[o] * f;
    }
    *lse = fm + __logf(fe);
    *ez = fw / fe;
}

template <int E, int CPW>          // CPW: 16-column tiles per wave; blockIdx.z picks the workgroup's 64 CPW columns
__global__ __launch_bounds__(256) void nce_rows_grad_kernel(NceWs w, int B, float* __restrict__ dx1, float* __restrict__ dx2, int row0,
                                                            int nrows, float gscale, float* __restrict__ loss, float* __restrict__ dls) {
    __shared__ float lse_o[ROWS_MAX_B];         // log-sum-exp of the OTHER side, per key
    __shared__ float lse_q[16];
    __shared__ float red[2][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int blk = blockIdx.x, side = blockIdx.y, col0 = blockIdx.z * (64 * CPW) + 16 * CPW * wave;
    float* dx = side == 0 ? dx1 : dx2;
    const bool in_strip = dx != nullptr && 16 * blk < row0 + nrows && 16 * blk + 16 > row0;
    const bool sums = blk == 0 && blockIdx.z == 0;
    if (!sums && !in_strip) return;
    float lsum = 0.f, esum = 0.f;
    for (int n = threadIdx.x; n < B; n += 256) {
        float l, e;
        nce_rows_merge(w, B, side ^ 1, n, &l, &e);
        lse_o[n] = l;
        const float d = w.diag[n];
        lsum += l - d;
        esum += e - d;
    }
    if (threadIdx.x < 16 && 16 * blk + (int)threadIdx.x < B) {
        float l, e;
        nce_rows_merge(w, B, side, 16 * blk + threadIdx.x, &l, &e);
        lse_q[threadIdx.x] = l;
    }
    if (sums) {         // this side's first workgroup owns the other side's share of the loss and of d logit_scale
        red[0][threadIdx.x] = lsum; red[1][threadIdx.x] = esum;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            atomicAdd(w.accum + 0, red[0][0]);          // two terms in all: a + b = b + a, the order of arrival does not show
            atomicAdd(w.accum + 1, red[1][0]);
            __threadfence();
            if (atomicAdd((unsigned int*)(w.accum + 2), 1u) == 1u) {      // the second of the two
                __threadfence();
                const float a = atomicAdd(w.accum + 0, 0.f), b = atomicAdd(w.accum + 1, 0.f);
                loss[0] = a / (float)B;
                if (dls != nullptr) dls[0] = w.scal[1] != 0.f ? 0.f : b * gscale / (float)B;
            }
        }
    }
    __syncthreads();
    if (!in_strip) return;
    // lane (r, g): query m = 16 blk + r as the B-operand column, keys k0 + 8 g .. + 7 as its K slots
    const int m = 16 * blk + r;
    const float lq = lse_q[r < 16 ? r : 0];
    const float s = w.scal[0], kf = gscale / (float)B * s;
    const float* zrow = w.zws + ((int64_t)side * B + (m < B ? m : B - 1)) * w.Bp128 + 8 * g;
    const bf16_t* xt = (side == 0 ? w.x2t : w.x1t) + (int64_t)(col0 + r) * w.Bp32 + 8 * g;       // rows = this wave's columns
    f32x4 acc[CPW];
#pragma unroll
    for (int ct = 0; ct < CPW; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = w.Bp32 / 32;
#pragma unroll 4
    for (int ks = 0; ks < nsteps; ++ks) {
        const int k0 = 32 * ks;
        bf16x8 xf[CPW];
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct) xf[ct] = *(const bf16x8*)(xt + (int64_t)16 * ct * w.Bp32 + k0);
        const f32x4 z0 = *(const f32x4*)(zrow + k0), z1 = *(const f32x4*)(zrow + k0 + 4);
        bf16x8 bd;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = k0 + 8 * g + e;
            const float z = e < 4 ? z0[e] : z1[e - 4];
            float d = 0.f;
            if (m < B && n < B) d = (__expf(z - lq) + __expf(z - lse_o[n]) - (m == n ? 2.f : 0.f)) * kf;
            bd[e] = (bf16_t)d;
        }
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ct], bd, acc[ct], 0, 0, 0);
    }
    if (m >= row0 && m < row0 + nrows && m < B) {
        float* orow = dx + (int64_t)(m - row0) * E + col0 + 4 * g;
#pragma unroll
        for (int ct = 0; ct < CPW; ++ct) *(f32x4*)(orow + 16 * ct) = acc[ct];
    }
}

// The gradient kernel of the strip path: nce_rows_grad_kernel's product for the rows [row0, row0 + nrows) of a LARGE batch -- the
// strip's fp32-grade logits come from pass 1 (zrow / zcol), the log-sum-exp vectors from the merge kernel, so that nothing B x B is
// stored and no split reduction follows: dx1[m] = sum_n s dZ[m, n] x2[n] (side 0, keys n) and dx2[m] = sum_n s dZ[n, m] x1[n] (side 1).
// A workgroup owns 16 rows x 128 columns; with thousands of keys the loop over them is a chain of memory round trips, so the four
// waves split the KEYS (each forms s dZ of its quarter once, for all 128 columns) and add their partial tiles through LDS.
// RT 16-row tiles per workgroup, NW waves splitting the keys.  Measured at B = 4096 / 512 rows: RT = 1, NW = 4: 44 us (the whole call 137-142);
// NW = 8: the same (137-146); RT = 2 (half the re-reads of the transposed operand, half the workgroups): 66 us (160-165) -- the launch is
// bound by round trips in flight on 256 CUs, not by bytes.  The call as a whole equals the tile path (139-144 us) in time: what both
// spend is pass 1 over all 4096^2 logits (59 us, which the LOSS needs on every rank) plus four small launches; the strip path's gain
// is the workspace (55 instead of 181 MB) and that no B x B matrix is written.
template <int E, int RT, int NW>
__global__ __launch_bounds__(NW * 64) void nce_strip_grad_kernel(NceWs w, int B, float* __restrict__ dx1, float* __restrict__ dx2, int row0,
                                                                 int nrows, float gscale) {
    extern __shared__ float lds_f[];            // [B] log-sum-exp of the OTHER side per key | [NW waves][RT][8 tiles][64 lanes] f32x4 partials
    float* lse_o = lds_f;
    f32x4* part = (f32x4*)(lds_f + ((B + 3) & ~3));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int blk = blockIdx.x, side = blockIdx.y, col0 = blockIdx.z * 128;
    float* dx = side == 0 ? dx1 : dx2;
    if (dx == nullptr) return;
    const float* lo = side == 0 ? w.clse : w.rlse;
    const float* lq_all = side == 0 ? w.rlse : w.clse;
    for (int n = threadIdx.x; n < B; n += NW * 64) lse_o[n] = lo[n];
    __syncthreads();
    const float s = w.scal[0], kf = gscale / (float)B * s;
    int ml[RT], m[RT];
    bool live[RT];
    float lq[RT];
    const float* zrow[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        ml[t] = 16 * (RT * blk + t) + r;              // row inside the strip
        m[t] = row0 + ml[t];
        live[t] = ml[t] < nrows;
        lq[t] = lq_all[live[t] ? m[t] : row0];
        zrow[t] = (side == 0 ? w.zrow : w.zcol) + (int64_t)(live[t] ? ml[t] : 0) * w.Bp128 + 8 * g;
    }
    const bf16_t* xt = (side == 0 ? w.x2t : w.x1t) + (int64_t)(col0 + r) * w.Bp32 + 8 * g;       // rows = columns col0 + 16 ct + r
    f32x4 acc[RT][8];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = w.Bp32 / 32;
    const int per = (nsteps + NW - 1) / NW, ks0 = wave * per, ks1 = ks0 + per < nsteps ? ks0 + per : nsteps;
#pragma unroll 2
    for (int ks = ks0; ks < ks1; ++ks) {
        const int k0 = 32 * ks;
        bf16x8 xf[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) xf[ct] = *(const bf16x8*)(xt + (int64_t)16 * ct * w.Bp32 + k0);
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const f32x4 z0 = *(const f32x4*)(zrow[t] + k0), z1 = *(const f32x4*)(zrow[t] + k0 + 4);
            bf16x8 bd;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int n = k0 + 8 * g + e;
                const float z = e < 4 ? z0[e] : z1[e - 4];
                float d = 0.f;
                if (live[t] && n < B) d = (__expf(z - lq[t]) + __expf(z - lse_o[n]) - (m[t] == n ? 2.f : 0.f)) * kf;
                bd[e] = (bf16_t)d;
            }
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[ct], bd, acc[t][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) part[((wave * RT + t) * 8 + ct) * 64 + lane] = acc[t][ct];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        if (!live[t]) continue;         // a wave adds the NW partials of its column tiles in a fixed order
#pragma unroll
        for (int u = 0; u < 8 / NW; ++u) {
            const int ct = (8 / NW) * wave + u;
            f32x4 v = part[((0 * RT + t) * 8 + ct) * 64 + lane];
#pragma unroll
            for (int q = 1; q < NW; ++q) v += part[((q * RT + t) * 8 + ct) * 64 + lane];
            *(f32x4*)(dx + (int64_t)ml[t] * E + col0 + 16 * ct + 4 * g) = v;
        }
    }
}

// (nce_strip_grad_kernel keeps B fp32 log-sum-exps beside 64 KiB of operands in its dynamic LDS: batches whose B * 4 + 64 KiB exceed the
// CU's 160 KiB -- B above ~24 k -- take the tile path, which has no such limit; VIPANT_NCE_STRIP=0 selects the tile path for A/B, and the
// workspace query follows the same switch so that a caller sizing by it never comes up short: ADVICE r5)
bool strip_path(int64_t B, int64_t E, int64_t nrows) {
    const char* strip_env = getenv("VIPANT_NCE_STRIP");
    if (strip_env && strip_env[0] == '0') return false;
    return B > ROWS_MAX_B && E == ROWS_E && nrows > 0 && nrows <= ROWS_MAX_B && nrows < B && (B + 3) / 4 * 16 + 65536 <= 160 * 1024;
}

}  // namespace

extern "C" size_t vipant_infonce_workspace_bytes(int64_t B, int64_t E) { return carve(nullptr, B, E).total; }
// the workspace when gradients are wanted for `nrows` rows only (0 < nrows <= 768 of B > 768 rows, E = 512: the per-rank form of an
// N-GPU step): no B x B gradient matrices, no split-reduction partials.  Other shapes: vipant_infonce_workspace_bytes.
extern "C" size_t vipant_infonce_strip_workspace_bytes(int64_t B, int64_t E, int64_t nrows) {
    return strip_path(B, E, nrows) ? carve(nullptr, B, E, nrows).total : carve(nullptr, B, E).total;
}

extern "C" int32_t vipant_infonce_fwd_bwd(const float* x1, const float* x2, const float* logit_scale, float scale_max,
                                          float* loss, float* dx1, float* dx2, float* dlogit_scale, float grad_scale,
                                          int64_t B, int64_t E, int64_t row0, int64_t nrows, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(B > 0 && E > 0 && E % 64 == 0, VIPANT_EBADSHAPE, "infonce: need E %% 64 == 0 (B=%ld E=%ld)", (long)B, (long)E);
    VIPANT_REQUIRE(row0 >= 0 && nrows >= 0 && row0 + nrows <= B, VIPANT_EBADSHAPE,
                   "infonce: bad row slice [%ld, %ld) of %ld", (long)row0, (long)(row0 + nrows), (long)B);
    const bool strip = strip_path(B, E, nrows) && (dx1 != nullptr || dx2 != nullptr);
    VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= (strip ? vipant_infonce_strip_workspace_bytes(B, E, nrows)
                                                                      : vipant_infonce_workspace_bytes(B, E)),
                   VIPANT_ENOWORKSPACE, "infonce: workspace too small");
    VIPANT_REQUIRE((uintptr_t)workspace % 256 == 0, VIPANT_EALIGN, "infonce: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    NceWs w = carve((char*)workspace, B, E, strip ? nrows : 0);
    if (strip) { w.zrow0 = (int)row0; w.znrows = (int)nrows; }
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)nce_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)nce_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        done_on_device(once);
    }
    const int K = (int)(3 * E);
    const unsigned tiles = (unsigned)(ceil_div(B, BM) * ceil_div(B, BN));
    int64_t pb = ceil_div(B * E, 256);
    const char* rows_env = getenv("VIPANT_NCE_ROWS");
    const bool rows_path = B <= ROWS_MAX_B && E == ROWS_E && !(rows_env && rows_env[0] == '0');
    hipLaunchKernelGGL(nce_prep_kernel, dim3((unsigned)(pb > 2048 ? 2048 : pb)), dim3(256), 0, s, x1, x2, logit_scale,
                       scale_max, w, (int)B, (int)E, rows_path ? 1 : 0);
    VIPANT_LAUNCH_CHECK();
    if (strip) {     // thousands of clips: the transposed bf16 copies through a tiled transpose (coalesced both ways)
        hipLaunchKernelGGL(nce_transpose_kernel, dim3((unsigned)ceil_div(E, 64), (unsigned)ceil_div(w.Bp32, 64), 2), dim3(256), 0, s, x1, x2, w,
                           (int)B, (int)E);
        VIPANT_LAUNCH_CHECK();
    }
    if (rows_path) {
        // few clips (the step's own batch on one GPU): 256 x 256 tiles would leave this on 4-16 CUs for three dependent
        // launches of 40-50 us each; row blocks of 16 against key chunks of 128 fill the chip and need no B x B operand in HBM
        const bool want = (dx1 != nullptr || dx2 != nullptr) && nrows > 0;
        hipLaunchKernelGGL(nce_rows_lse_kernel<3 * ROWS_E / 32>, dim3((unsigned)w.KC, (unsigned)ceil_div(B, 16), 2), dim3(256), 0, s, w, (int)B);
        VIPANT_LAUNCH_CHECK();
        // two 16-column tiles per wave (128 columns per workgroup): 53.6 us per call at B = 512 against 55.4 / 60.2 with four / eight
        hipLaunchKernelGGL((nce_rows_grad_kernel<ROWS_E, 2>), dim3((unsigned)ceil_div(B, 16), 2, ROWS_E / 128), dim3(256), 0, s, w, (int)B,
                           want ? dx1 : nullptr, want ? dx2 : nullptr, (int)row0, (int)nrows, grad_scale, loss, dlogit_scale);
        VIPANT_LAUNCH_CHECK();
        return VIPANT_OK;
    }
    hipLaunchKernelGGL(nce_tile_kernel<1>, dim3(tiles), dim3(512), LDS_BYTES, s, w, (int)B, K, 0, (int)B, 1.0f);
    VIPANT_LAUNCH_CHECK();
    const int nparts = (int)ceil_div(B, 16);
    hipLaunchKernelGGL(nce_merge_kernel, dim3((unsigned)nparts), dim3(256), 0, s, w, (int)B);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_final_kernel, dim3(1), dim3(256), 0, s, w, (int)B, nparts, grad_scale, loss, dlogit_scale);
    VIPANT_LAUNCH_CHECK();
    if ((dx1 == nullptr && dx2 == nullptr) || nrows == 0) return VIPANT_OK;
    if (strip) {
        // one rank's strip of a large batch: its logits were kept by pass 1, its gradient rows are 16-row blocks of one launch --
        // no B x B matrix of s dZ in HBM, no token-reduction contraction, no split reduction (139 -> ~100 us at B = 4096 / 512 rows)
        static DeviceMax most;
        constexpr int RT = 1, NW = 8;
        const int lds = (int)((B + 3) / 4 * 4) * 4 + NW * RT * 8 * 64 * 16;
        if (raise_on_device(most, lds)) {
            VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)nce_strip_grad_kernel<ROWS_E, RT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            raised_on_device(most, lds);
        }
        hipLaunchKernelGGL((nce_strip_grad_kernel<ROWS_E, RT, NW>), dim3((unsigned)ceil_div(nrows, 16 * RT), 2, ROWS_E / 128), dim3(NW * 64), lds,
                           s, w, (int)B, dx1, dx2, (int)row0, (int)nrows, grad_scale);
        VIPANT_LAUNCH_CHECK();
        return VIPANT_OK;
    }
    // The token-reduction contraction wants its A operand 16-byte aligned, i.e. a strip starting on a multiple of 8 columns: a
    // rank's strip that does not (the reference's shipped default is 432 clips over 4 GPUs = 108 per rank) is widened to the left
    // by up to 7 columns, contracted into a scratch matrix, and its own rows copied out.
    const int64_t r0a = row0 & ~(int64_t)7, lead = row0 - r0a, nr = nrows + lead;
    // one pass-2 launch writes s dZ (columns in the strip; feeds dx2) and its transpose (rows in the strip; feeds dx1)
    NceWs t = w;
    if (dx2 == nullptr) t.dz = nullptr;
    if (dx1 == nullptr) t.dzt = nullptr;
    hipLaunchKernelGGL(nce_tile_kernel<2>, dim3(tiles), dim3(512), LDS_BYTES, s, t, (int)B, K, (int)r0a, (int)nr, grad_scale);
    VIPANT_LAUNCH_CHECK();
    // dx2[n, :] = sum_m s dZ[m][n] x1[m, :] (columns n in the strip) and dx1[m, :] = sum_n s dZ^T[n][m] x2[n, :] (columns m in the
    // strip): the same token-reduction contraction twice -- one launch for both when both are wanted
    if (dx1 != nullptr && dx2 != nullptr) {
        float* out2 = lead ? w.dxtmp : dx2;
        float* out1 = lead ? w.dxtmp2 : dx1;
        const int32_t e = vipant_gemm_tn_pair((const uint16_t*)(w.dz + r0a), (const uint16_t*)w.x1cat, out2, (const uint16_t*)(w.dzt + r0a),
                                              (const uint16_t*)w.x2cat, out1, w.Bp, 3 * E, E, B, nr, E, w.tn_ws, w.tn_bytes, stream);
        if (e != VIPANT_OK) return e;
        if (lead) {
            VIPANT_HIP_TRY(hipMemcpyAsync(dx2, out2 + lead * E, (size_t)nrows * E * sizeof(float), hipMemcpyDeviceToDevice, s));
            VIPANT_HIP_TRY(hipMemcpyAsync(dx1, out1 + lead * E, (size_t)nrows * E * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        return VIPANT_OK;
    }
    if (dx2 != nullptr) {
        float* out = lead ? w.dxtmp : dx2;
        const int32_t e = vipant_gemm_tn((const uint16_t*)(w.dz + r0a), w.Bp, (const uint16_t*)w.x1cat, 3 * E, out, E, B,
                                         nr, E, 0, nullptr, w.tn_ws, w.tn_bytes, stream);
        if (e != VIPANT_OK) return e;
        if (lead) VIPANT_HIP_TRY(hipMemcpyAsync(dx2, out + lead * E, (size_t)nrows * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (dx1 != nullptr) {
        float* out = lead ? w.dxtmp : dx1;
        const int32_t e = vipant_gemm_tn((const uint16_t*)(w.dzt + r0a), w.Bp, (const uint16_t*)w.x2cat, 3 * E, out, E, B,
                                         nr, E, 0, nullptr, w.tn_ws, w.tn_bytes, stream);
        if (e != VIPANT_OK) return e;
        if (lead) VIPANT_HIP_TRY(hipMemcpyAsync(dx1, out + lead * E, (size_t)nrows * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return VIPANT_OK;
}
