// HBM-bound layout / elementwise kernels around the contractions: casts (with transpose), patch im2col,
// token assembly (cls + positional table), L2 normalisation, token-embedding gather, column sums.
// All use 16-byte accesses per lane on the contiguous dimension and grid-stride loops capped at ~2048
// workgroups (MI355X: 256 CUs x 8).
#include "common.h"

namespace {

constexpr int MAXB = 2048;
inline int grid_for(int64_t work_items, int per_block) {
    int64_t b = ceil_div(work_items, per_block);
    return (int)(b > MAXB ? MAXB : (b < 1 ? 1 : b));
}

// ---- fp32 -> bf16 cast, optional transposed copy (64x64 tiles through LDS) --------------------------------
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        *(bf16x4*)(dst + i * 4) = f32x4_to_bf16x4(*(const f32x4*)(src + i * 4));
}

__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             bf16_t* __restrict__ dst_t, int R, int C) {
    __shared__ bf16_t tile[64][66];
    const int tr = blockIdx.y * 64, tc = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int gr = tr + r, gc = tc + tx;
        bf16_t v = (bf16_t)0.0f;
        if (gr < R && gc < C) {
            v = (bf16_t)src[(int64_t)gr * C + gc];
            if (dst != nullptr) dst[(int64_t)gr * C + gc] = v;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int gc = tc + c, gr = tr + tx;
        if (gc < C && gr < R) dst_t[(int64_t)gc * R + gr] = tile[tx][c];
    }
}

// All the weight matrices of a tower in ONE launch (48 small launches of the kernel above cost ~1 ms per step): workgroup ->
// (tensor, 64x64 tile) through a prefix table of tile counts; 16-B loads, 8-B stores of the straight copy, the transposed copy
// through LDS.
struct CastMulti {
    const float* const* src; bf16_t* const* dst; bf16_t* const* dst_t;
    const int32_t* R; const int32_t* C; const int32_t* tile_start;      // tile_start[n + 1]
    int n;
};
__global__ __launch_bounds__(256) void cast_multi_kernel(CastMulti a) {
    __shared__ bf16_t tile[64][68];
    int lo = 0, hi = a.n;                        // largest t with tile_start[t] <= blockIdx.x
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.tile_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid;
    }
    const int t = lo;
    const int R = a.R[t], C = a.C[t];
    const int tcols = (C + 63) / 64;
    const int local = (int)blockIdx.x - a.tile_start[t];
    const int tr = (local / tcols) * 64, tc = (local % tcols) * 64;
    const float* src = a.src[t];
    bf16_t* dst = a.dst[t];
    bf16_t* dst_t = a.dst_t[t];
    const int cx = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;          // 16 threads x 4 columns per row, 16 rows per pass
    const bool vec = (C & 3) == 0;
#pragma unroll
    for (int r = ry; r < 64; r += 16) {
        const int gr = tr + r, gc = tc + cx;
        bf16x4 v = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        if (gr < R) {
            if (vec && gc + 3 < C) {
                v = f32x4_to_bf16x4(*(const f32x4*)(src + (int64_t)gr * C + gc));
                if (dst != nullptr) *(bf16x4*)(dst + (int64_t)gr * C + gc) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gc + e < C) {
                        v[e] = (bf16_t)src[(int64_t)gr * C + gc + e];
                        if (dst != nullptr) dst[(int64_t)gr * C + gc + e] = v[e];
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][cx + e] = v[e];
    }
    __syncthreads();
    if (dst_t == nullptr) return;
    const bool vect = (R & 3) == 0;
#pragma unroll
    for (int c = ry; c < 64; c += 16) {           // output row = source column tc + c, 4 consecutive source rows per thread
        const int gc = tc + c, gr = tr + cx;
        if (gc >= C) continue;
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[cx + e][c];
        if (vect && gr + 3 < R) {
            *(bf16x4*)(dst_t + (int64_t)gc * R + gr) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (gr + e < R) dst_t[(int64_t)gc * R + gr + e] = v[e];
        }
    }
}

template <typename XT>
__global__ __launch_bounds__(256) void residual_add_kernel(const XT* __restrict__ x, const bf16_t* __restrict__ add,
                                                           float* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const bf16x4 a = *(const bf16x4*)(add + i * 4);
        f32x4 v;
        if constexpr (sizeof(XT) == 4) {
            v = *(const f32x4*)(x + i * 4);
        } else {
            typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
            const f16x4_t h = *(const f16x4_t*)(x + i * 4);
            v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
        *(f32x4*)(out + i * 4) = v + f32x4{(float)a[0], (float)a[1], (float)a[2], (float)a[3]};
    }
}

// ---- conv1.weight -> GEMM weight ---------------------------------------------------------------------
__global__ void conv_weight_prep_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int64_t O, int Cin,
                                        int khw, int mean_channels) {
    const int64_t total = mean_channels ? O * khw : O * Cin * khw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (mean_channels) {
            const int64_t o = i / khw, k = i % khw;
            float s = 0.f;
            for (int c = 0; c < Cin; ++c) s += w[(o * Cin + c) * khw + k];
            out[i] = (bf16_t)(s / (float)Cin);
        } else {
            out[i] = (bf16_t)w[i];
        }
    }
}

__global__ void conv_weight_grad_kernel(const float* __restrict__ g, float* __restrict__ wgrad, int64_t O, int Cin,
                                        int khw, int accumulate) {
    const int64_t total = O * Cin * khw;
    const float inv = 1.0f / (float)Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = i / ((int64_t)Cin * khw), k = i % khw;
        const float v = g[o * khw + k] * inv;
        wgrad[i] = accumulate ? wgrad[i] + v : v;
    }
}

// ---- im2col: each thread emits 8 consecutive patch columns (16 B of bf16) ------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int b, int C,
                                                     int T, int F, int ph, int pw, int sh, int sw, int nrow, int ncol,
                                                     int vec_ok) {
    const int kcols = C * ph * pw;
    const int64_t chunks_per_row = kcols / 8;
    const int64_t total = (int64_t)b * nrow * ncol * chunks_per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t prow = i / chunks_per_row;
        const int k = (int)(i % chunks_per_row) * 8;
        const int c = k / (ph * pw), ii = (k / pw) % ph, jj = k % pw;
        const int64_t bi = prow / (nrow * ncol);
        const int tok = (int)(prow % (nrow * ncol));
        const int tr = tok / ncol, fc = tok % ncol;
        const float* src = x + ((bi * C + c) * T + (int64_t)tr * sh + ii) * F + fc * sw + jj;
        bf16x8 o;
        if (vec_ok) {
            const f32x4 a = *(const f32x4*)src, d = *(const f32x4*)(src + 4);
            o[0] = (bf16_t)a[0]; o[1] = (bf16_t)a[1]; o[2] = (bf16_t)a[2]; o[3] = (bf16_t)a[3];
            o[4] = (bf16_t)d[0]; o[5] = (bf16_t)d[1]; o[6] = (bf16_t)d[2]; o[7] = (bf16_t)d[3];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)src[e];
        }
        *(bf16x8*)(out + prow * kcols + k) = o;
    }
}

// ---- token assembly ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ patches, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, float* __restrict__ tokens, int64_t b,
                                                       int P, int D) {
    const int S = P + 1, d4 = D / 4;
    const int64_t total = b * S * d4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d4) * 4;
        const int64_t row = i / d4;
        const int s = (int)(row % S);
        const int64_t bi = row / S;
        const f32x4 base = s == 0 ? *(const f32x4*)(cls + c) : *(const f32x4*)(patches + (bi * P + s - 1) * D + c);
        *(f32x4*)(tokens + row * D + c) = base + *(const f32x4*)(pos + (int64_t)s * D + c);
    }
}

// tokens[item * S, :] = cls + pos[0, :]: the class-token rows (the patch rows come from vipant_gemm_nt_tokens)
__global__ __launch_bounds__(256) void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos,
                                                       float* __restrict__ tokens, int64_t b, int64_t S, int D) {
    const int d4 = D / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < b * d4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d4) * 4;
        *(f32x4*)(tokens + (i / d4) * S * D + c) = *(const f32x4*)(cls + c) + *(const f32x4*)(pos + c);
    }
}

// dpatches (bf16) = dtokens rows s >= 1; dpos[s] = sum_b dtokens[b, s]; dcls = dpos row 0 (before accumulate).
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dtok, bf16_t* __restrict__ dpatches,
                                                           float* dcls, float* dpos, int accumulate, int64_t b, int P,
                                                           int D) {
    const int S = P + 1, d4 = D / 4;
    const int64_t total = (int64_t)S * d4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d4) * 4;
        const int s = (int)(i / d4);
        // eight samples per trip: eight independent loads in flight per thread (the one-sample loop was a 512-deep chain of
        // dependent round trips: 382 us for 0.74 GB), summed in a fixed order (reproducible)
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        int64_t bi = 0;
        for (; bi + 8 <= b; bi += 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(dtok + ((bi + j) * S + s) * D + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (s > 0) *(bf16x4*)(dpatches + ((bi + j) * P + s - 1) * D + c) = f32x4_to_bf16x4(v[j]);
            }
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; bi < b; ++bi) {
            const f32x4 v = *(const f32x4*)(dtok + (bi * S + s) * D + c);
            acc += v;
            if (s > 0) *(bf16x4*)(dpatches + (bi * P + s - 1) * D + c) = f32x4_to_bf16x4(v);
        }
        float* dp = dpos + (int64_t)s * D + c;
        if (s == 0) {
            float* dc = dcls + c;
            *(f32x4*)dc = accumulate ? *(const f32x4*)dc + acc : acc;
        }
        *(f32x4*)dp = accumulate ? *(const f32x4*)dp + acc : acc;
    }
}

// ---- L2 normalisation: one wave per row --------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ norm, int64_t M, int E) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane * 4; c < E; c += 256) {
            const f32x4 v = *(const f32x4*)(x + row * E + c);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        const float n = sqrtf(wave_sum(s));
        const float inv = 1.0f / n;
        for (int c = lane * 4; c < E; c += 256) *(f32x4*)(y + row * E + c) = *(const f32x4*)(x + row * E + c) * inv;
        if (lane == 0) norm[row] = n;
    }
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ norm, float* __restrict__ dx,
                                                         bf16_t* __restrict__ dxb, int64_t M, int E) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane * 4; c < E; c += 256) {
            const f32x4 g = *(const f32x4*)(dy + row * E + c), v = *(const f32x4*)(y + row * E + c);
            s += g[0] * v[0] + g[1] * v[1] + g[2] * v[2] + g[3] * v[3];
        }
        s = wave_sum(s);
        const float inv = 1.0f / norm[row];
        for (int c = lane * 4; c < E; c += 256) {
            const f32x4 o = (*(const f32x4*)(dy + row * E + c) - *(const f32x4*)(y + row * E + c) * s) * inv;
            if (dx != nullptr) *(f32x4*)(dx + row * E + c) = o;
            if (dxb != nullptr) *(bf16x4*)(dxb + row * E + c) = f32x4_to_bf16x4(o);
        }
    }
}

// ---- text: embedding gather + positional table, EOT argmax ---------------------------------------------
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ table,
                                                           const float* __restrict__ pos, float* __restrict__ x,
                                                           int64_t* __restrict__ eot, int64_t b, int L, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = b * L;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int l = (int)(row % L);
        const int64_t tok = tokens[row];
        for (int c = lane * 4; c < D; c += 256)
            *(f32x4*)(x + row * D + c) = *(const f32x4*)(table + tok * D + c) + *(const f32x4*)(pos + (int64_t)l * D + c);
        if (l == 0 && lane == 0) {   // first maximal index, as torch.argmax
            int64_t best = tokens[row];
            int bi = 0;
            for (int j = 1; j < L; ++j)
                if (tokens[row + j] > best) { best = tokens[row + j]; bi = j; }
            eot[row / L] = bi;
        }
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t n, int64_t rpi, int D) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = i * rpi + (idx != nullptr ? (idx[i] < 0 ? 0 : (idx[i] >= rpi ? rpi - 1 : idx[i])) : 0);      // clamped into the item
        for (int c = lane * 4; c < D; c += 256) *(f32x4*)(out + i * D + c) = *(const f32x4*)(x + r * D + c);
    }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                                           float* __restrict__ dx, int64_t n, int64_t rpi, int D) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = i * rpi + (idx != nullptr ? (idx[i] < 0 ? 0 : (idx[i] >= rpi ? rpi - 1 : idx[i])) : 0);      // clamped into the item
        for (int c = lane * 4; c < D; c += 256) {
            float* d = dx + r * D + c;
            *(f32x4*)d = *(const f32x4*)d + *(const f32x4*)(g + i * D + c);
        }
    }
}

// compact fp32 rows g [n, D] -> bf16 rows r_i = i * rpi + idx[i] (idx == NULL: + 0) of a ZEROED bf16 matrix, and their bf16 copy
__global__ __launch_bounds__(256) void scatter_rows_bf16_kernel(const float* __restrict__ g, const int64_t* __restrict__ idx,
                                                                bf16_t* __restrict__ dx, bf16_t* __restrict__ compact, int64_t n,
                                                                int64_t rpi, int D) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = i * rpi + (idx != nullptr ? (idx[i] < 0 ? 0 : (idx[i] >= rpi ? rpi - 1 : idx[i])) : 0);      // clamped into the item
        for (int c = lane * 4; c < D; c += 256) {
            const bf16x4 v = f32x4_to_bf16x4(*(const f32x4*)(g + i * D + c));
            *(bf16x4*)(dx + r * D + c) = v;
            if (compact != nullptr) *(bf16x4*)(compact + i * D + c) = v;
        }
    }
}

// ---- column sums over tokens: thread = 8 columns, block = 512 columns x 4 row lanes, grid.y row chunks --
constexpr int CS_ROWCHUNKS = 128;
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, int64_t ldx, float* __restrict__ partial,
                                                     int64_t M, int N) {
    __shared__ float red[4][512];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 512 + tx * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col < N) {
        for (int64_t r = (int64_t)blockIdx.y * 4 + ty; r < M; r += (int64_t)gridDim.y * 4) {
            const bf16x8 v = *(const bf16x8*)(X + r * ldx + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[e];
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {
        const int c = blockIdx.x * 512 + i;
        if (c < N) partial[(int64_t)blockIdx.y * N + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

__global__ void colsum_finalize_kernel(const float* __restrict__ partial, int nchunks, int N, float* out, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float s = 0.f;
#pragma unroll 8          // eight loads in flight: this loop is a chain of global-load latencies otherwise (30 us for 128 chunks)
    for (int k = 0; k < nchunks; ++k) s += partial[(int64_t)k * N + i];
    out[i] = accumulate ? out[i] + s : s;
}

__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const bf16x4 v = *(const bf16x4*)(src + i * 4);
        *(f32x4*)(dst + i * 4) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
}


// bf16 rows -> e4m3 (OCP e4m3fn) rows with ONE power-of-two scale per row (BASELINE.json configs[4]: fp8 operands of the block
// contractions): e = the smallest exponent with amax / 2^e <= 448, q = rne(x / 2^e), scale byte = e + 127 (E8M0, the form the
// block-scale operand of v_mfma_scale_f32_16x16x128_f8f6f4 takes).  One wave per row, the row held in registers between finding its maximum and converting it.
template <int NV>      // NV = ceil(K / 512): 16-byte chunks per lane; the row stays in registers between the two passes over it
__global__ __launch_bounds__(256) void quant_e4m3_rows_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                              int64_t ldq, uint8_t* __restrict__ scale, int64_t M, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + row * ldx;
    bf16x8 v[NV];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 8 + i * 512;
        v[i] = c < K ? *(const bf16x8*)(xr + c) : bf16x8{};
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf((float)v[i][e]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    int e = 0;
    if (amax > 0.f) {
        e = (int)((__float_as_uint(amax) >> 23) & 255u) - 127 - 8;            // floor(log2(amax)) - 8: amax / 2^e in [256, 512)
        if (amax * __uint_as_float((uint32_t)(127 - e) << 23) > 448.f) e += 1;
        e = e < -127 ? -127 : (e > 127 ? 127 : e);
    }
    const float inv = __uint_as_float((uint32_t)(127 - e) << 23);              // 2^-e
    if (lane == 0) scale[row] = (uint8_t)(e + 127);
    uint8_t* qr = q + row * ldq;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane * 8 + i * 512;
        if (c < K) {
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][0] * inv, (float)v[i][1] * inv, w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][2] * inv, (float)v[i][3] * inv, w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][4] * inv, (float)v[i][5] * inv, w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32((float)v[i][6] * inv, (float)v[i][7] * inv, w1, true);
            *(int2*)(qr + c) = int2{w0, w1};
        }
    }
}

// bf16 rows -> e4m3 with one power-of-two scale per 32 consecutive elements of a row (the MX block format, common.h): a thread takes 8
// elements, the four threads of a block agree on the exponent through DPP.  Purely streaming: no row-wide reduction.
__global__ __launch_bounds__(256) void quant_e4m3_mx_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                            int64_t ldq, uint8_t* __restrict__ scale, int64_t M, int K, int kt_row,
                                                            int kb0) {
    const int cpr = K / 8;                               // 16-byte chunks per row (a multiple of 4)
    const int64_t total = M * cpr;
    const int64_t span = (int64_t)gridDim.x * 256;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += span) {
        const int64_t c = base + threadIdx.x;             // whole quads are in or out: total % 4 == 0 and base % 4 == 0
        if (c >= total) break;
        const int64_t row = c / cpr;
        const int col = (int)(c - row * cpr) * 8;
        const u32x4 w = *(const u32x4*)(x + row * ldx + col);                // eight bf16
        float sc;
        const uint32_t e = mx_scale_byte<4>(mx_absmax2(mx_absmax2(mx_absmax2(mx_absmax2(0u, w[0]), w[1]), w[2]), w[3]), &sc) - 127u;
        *(int2*)(q + row * ldq + col) = int2{mx_pack4_bf16(w[0], w[1], sc), mx_pack4_bf16(w[2], w[3], sc)};
        if ((threadIdx.x & 3) == 0) scale[mx_scale_offset(row, kb0 + (col >> 5), kt_row)] = (uint8_t)(e + 127);
    }
}


// Block-uniform MX scales (the operand format of vipant_gemm_tn_e4m3): one power-of-two scale per aligned block of 32 ROWS x 32 columns,
// stored -- redundantly, 32 times -- in the MX layout of the row-wise format, so that the NT contractions read such a matrix as they
// read any other (a block scale is a valid scale for every one of its rows) and the weight-gradient contraction, whose k runs along
// the rows, finds one scale per 32 k of a column.  A workgroup takes 32 rows x 256 columns: thread t holds the 16-byte chunk t & 31 of
// rows (t >> 5) + 8 i; the block maximum goes through DPP (the four chunks of a 32-column block are four lanes) and an 8 x 8 LDS table.
__global__ __launch_bounds__(256) void quant_e4m3_mx32_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                              int64_t ldq, uint8_t* __restrict__ scale, int64_t M, int K, int nct,
                                                              int kt_row, int kb0) {
    __shared__ uint32_t tab[8][8];
    const int t = threadIdx.x, ch = t & 31, rg = t >> 5;
    const int64_t nrt = (M + 31) / 32;
    for (int64_t tile = blockIdx.x; tile < nrt * nct; tile += gridDim.x) {
        const int64_t r0 = (tile / nct) * 32;
        const int col = (int)(tile % nct) * 256 + ch * 8;
        const bool col_ok = col < K;                          // whole 32-column blocks are in or out (K % 32 == 0)
        u32x4 w[4];
        uint32_t mx = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = r0 + rg + 8 * i;
            w[i] = (col_ok && row < M) ? *(const u32x4*)(x + row * ldx + col) : u32x4{0u, 0u, 0u, 0u};
            mx = mx_absmax2(mx_absmax2(mx_absmax2(mx_absmax2(mx, w[i][0]), w[i][1]), w[i][2]), w[i][3]);
        }
        uint32_t m1 = max(mx & 0xFFFFu, mx >> 16);
        m1 = mx_lane_max_u<4>(m1);
        if ((ch & 3) == 0) tab[rg][ch >> 2] = m1;
        __syncthreads();
        uint32_t bm = 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) bm = max(bm, tab[r][ch >> 2]);
        float sc;
        const uint32_t byte = mx_scale_of_max(bm, &sc);
        if (col_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = r0 + rg + 8 * i;
                if (row < M) *(int2*)(q + row * ldq + col) = int2{mx_pack4_bf16(w[i][0], w[i][1], sc), mx_pack4_bf16(w[i][2], w[i][3], sc)};
            }
        }
        // the 32 rows' scale bytes of the tile's 8 column blocks: thread t writes (row t >> 3, block t & 7)
        {
            uint32_t bm2 = 0u;
#pragma unroll
            for (int r = 0; r < 8; ++r) bm2 = max(bm2, tab[r][t & 7]);
            float unused;
            const uint32_t b2 = mx_scale_of_max(bm2, &unused);
            const int64_t row = r0 + (t >> 3);
            const int kb = (int)(tile % nct) * 8 + (t & 7);
            if (row < M && kb * 32 < K) scale[mx_scale_offset(row, kb0 + kb, kt_row)] = (uint8_t)b2;
        }
        (void)byte;
        __syncthreads();
    }
}

// In place: an e4m3 matrix with row-wise MX scales (what the producers emit) -> block-uniform scales.  The block's scale is the
// largest of its 32 rows' scales; a row whose scale was smaller by d has its bytes divided by 2^d -- exact (a power of two) unless
// the result falls below e4m3's normal range (values 2^15 below the block's largest, rounded to the subnormal grid).  Rows that
// already carry the block's scale are not rewritten.
__global__ __launch_bounds__(256) void mx_uniform32_kernel(uint8_t* __restrict__ q, int64_t ldq, uint8_t* __restrict__ scale, int64_t M,
                                                           int K, int nct, int ktr, int kb0) {
    __shared__ uint32_t tab[8][8];
    const int t = threadIdx.x, ch = t & 31, rg = t >> 5;
    const int64_t nrt = (M + 31) / 32;
    for (int64_t tile = blockIdx.x; tile < nrt * nct; tile += gridDim.x) {
        const int64_t r0 = (tile / nct) * 32;
        const int col = (int)(tile % nct) * 256 + ch * 8;
        const int kb = kb0 + (col >> 5);
        const bool col_ok = col < K;
        uint32_t sb[4], smax = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = r0 + rg + 8 * i;
            sb[i] = (col_ok && row < M) ? scale[mx_scale_offset(row, kb, ktr)] : 0u;
            smax = max(smax, sb[i]);
        }
        if ((ch & 3) == 0) tab[rg][ch >> 2] = smax;
        __syncthreads();
        uint32_t S = 0u;
#pragma unroll
        for (int r = 0; r < 8; ++r) S = max(S, tab[r][ch >> 2]);
        if (col_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = r0 + rg + 8 * i;
                if (row >= M || sb[i] == S) continue;
                const uint32_t d = S - sb[i];
                const float f = d >= 64u ? 0.f : __uint_as_float((127u - d) << 23);       // 2^-d
                uint8_t* pq = q + row * ldq + col;
                const int2 v = *(const int2*)pq;
                int o[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int src = h ? v.y : v.x;
                    const auto lo = __builtin_amdgcn_cvt_pk_f32_fp8(src, false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8(src, true);
                    int w = 0;
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0] * f, lo[1] * f, w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(hi2[0] * f, hi2[1] * f, w, true);
                    o[h] = w;
                }
                *(int2*)pq = int2{o[0], o[1]};
                if ((ch & 3) == 0) scale[mx_scale_offset(row, kb, ktr)] = (uint8_t)S;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t vipant_mx_scale_bytes(int64_t M, int64_t K) { return K > 0 && K % 128 == 0 && M > 0 ? mx_scale_bytes(M, K) : 0; }

// K columns starting at column 32 kb0 of rows that are 128 kt_row elements long in the scale layout (x, q point at the first of them)
extern "C" int32_t vipant_quant_e4m3_mx_cols(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                                             int64_t kt_row, int64_t kb0, void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && kb0 >= 0 && kb0 * 32 + K <= kt_row * 128, VIPANT_EBADSHAPE,
                   "quant_e4m3_mx: need K %% 32 == 0 inside a row of 128 kt_row elements (M=%ld K=%ld kt_row=%ld kb0=%ld)", (long)M, (long)K,
                   (long)kt_row, (long)kb0);
    VIPANT_REQUIRE(ldx >= K && ldq >= K && ldx % 8 == 0 && ldq % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)q % 8 == 0 && scale != nullptr,
                   VIPANT_EALIGN, "quant_e4m3_mx: misaligned rows");
    const int64_t blocks = ceil_div(M * (K / 8), 256);
    hipLaunchKernelGGL(quant_e4m3_mx_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, q, ldq, scale, M, (int)K, (int)kt_row, (int)kb0);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_quant_e4m3_mx(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                                        void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 128 == 0, VIPANT_EBADSHAPE, "quant_e4m3_mx: need K %% 128 == 0 (M=%ld K=%ld)", (long)M, (long)K);
    return vipant_quant_e4m3_mx_cols(x, ldx, q, ldq, scale, M, K, K / 128, 0, stream);
}

// K columns (K % 32 == 0) starting at column 32 kb0 of rows that are 128 kt_row elements long in the scale layout (x, q point at the first)
extern "C" int32_t vipant_quant_e4m3_mx32_cols(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                                               int64_t kt_row, int64_t kb0, void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && kb0 >= 0 && kb0 * 32 + K <= kt_row * 128, VIPANT_EBADSHAPE,
                   "quant_e4m3_mx32: need K %% 32 == 0 inside a row of 128 kt_row elements (M=%ld K=%ld kt_row=%ld kb0=%ld)", (long)M, (long)K,
                   (long)kt_row, (long)kb0);
    VIPANT_REQUIRE(ldx >= K && ldq >= K && ldx % 8 == 0 && ldq % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)q % 8 == 0 && scale != nullptr,
                   VIPANT_EALIGN, "quant_e4m3_mx32: misaligned rows");
    const int64_t nct = ceil_div(K, 256), tiles = ceil_div(M, 32) * nct;
    hipLaunchKernelGGL(quant_e4m3_mx32_kernel, dim3((unsigned)(tiles < 8192 ? tiles : 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, q, ldq, scale, M, (int)K, (int)nct, (int)kt_row, (int)kb0);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_quant_e4m3_mx32(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K,
                                          void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 128 == 0, VIPANT_EBADSHAPE, "quant_e4m3_mx32: need K %% 128 == 0 (M=%ld K=%ld)", (long)M, (long)K);
    return vipant_quant_e4m3_mx32_cols(x, ldx, q, ldq, scale, M, K, K / 128, 0, stream);
}

extern "C" int32_t vipant_mx_uniform32_cols(uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K, int64_t kt_row, int64_t kb0,
                                            void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 32 == 0 && kb0 >= 0 && kb0 * 32 + K <= kt_row * 128, VIPANT_EBADSHAPE,
                   "mx_uniform32: need K %% 32 == 0 inside a row of 128 kt_row elements (M=%ld K=%ld kt_row=%ld kb0=%ld)", (long)M, (long)K,
                   (long)kt_row, (long)kb0);
    VIPANT_REQUIRE(ldq >= K && ldq % 8 == 0 && (uintptr_t)q % 8 == 0 && scale != nullptr, VIPANT_EALIGN, "mx_uniform32: misaligned rows");
    const int64_t nct = ceil_div(K, 256), tiles = ceil_div(M, 32) * nct;
    hipLaunchKernelGGL(mx_uniform32_kernel, dim3((unsigned)(tiles < 8192 ? tiles : 8192)), dim3(256), 0, (hipStream_t)stream, q, ldq, scale,
                       M, (int)K, (int)nct, (int)kt_row, (int)kb0);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_mx_uniform32(uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M, int64_t K, void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 128 == 0, VIPANT_EBADSHAPE, "mx_uniform32: need K %% 128 == 0 (M=%ld K=%ld)", (long)M, (long)K);
    return vipant_mx_uniform32_cols(q, ldq, scale, M, K, K / 128, 0, stream);
}

extern "C" int32_t vipant_quant_e4m3_rows(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scale, int64_t M,
                                          int64_t K, void* stream) {
    VIPANT_REQUIRE(M > 0 && K > 0 && K % 8 == 0 && K <= 8192, VIPANT_EBADSHAPE, "quant_e4m3_rows: need K %% 8 == 0 and K <= 8192 (M=%ld K=%ld)",
                   (long)M, (long)K);
    VIPANT_REQUIRE(ldx >= K && ldq >= K && ldx % 8 == 0 && ldq % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)q % 8 == 0, VIPANT_EALIGN,
                   "quant_e4m3_rows: misaligned rows");
    const dim3 grid((unsigned)ceil_div(M, 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define VIPANT_QUANT_CASE(NV) \
    case NV: hipLaunchKernelGGL(quant_e4m3_rows_kernel<NV>, grid, block, 0, st, (const bf16_t*)x, ldx, q, ldq, scale, M, (int)K); break;
    switch ((int)ceil_div(K, 512)) {
        VIPANT_QUANT_CASE(1) VIPANT_QUANT_CASE(2) VIPANT_QUANT_CASE(3) VIPANT_QUANT_CASE(4) VIPANT_QUANT_CASE(5) VIPANT_QUANT_CASE(6)
        VIPANT_QUANT_CASE(7) VIPANT_QUANT_CASE(8) VIPANT_QUANT_CASE(9) VIPANT_QUANT_CASE(10) VIPANT_QUANT_CASE(11) VIPANT_QUANT_CASE(12)
        VIPANT_QUANT_CASE(13) VIPANT_QUANT_CASE(14) VIPANT_QUANT_CASE(15) VIPANT_QUANT_CASE(16)
        default:
            vipant_set_error("quant_e4m3_rows: rows longer than 8192 elements are not built (K=%ld)", (long)K);
            return VIPANT_EBADSHAPE;
    }
#undef VIPANT_QUANT_CASE
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_cast_f32(const uint16_t* src, float* dst, int64_t n, void* stream) {
    VIPANT_REQUIRE(n > 0 && n % 4 == 0, VIPANT_EBADSHAPE, "cast_f32: n must be a positive multiple of 4 (n=%ld)", (long)n);
    VIPANT_REQUIRE((uintptr_t)src % 8 == 0 && (uintptr_t)dst % 16 == 0, VIPANT_EALIGN, "cast_f32: misaligned");
    int64_t blocks = ceil_div(n / 4, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n / 4);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_cast_bf16(const float* src, uint16_t* dst, uint16_t* dst_t, int64_t R, int64_t C, void* stream) {
    VIPANT_REQUIRE(R > 0 && C > 0, VIPANT_EBADSHAPE, "cast_bf16: empty");
    hipStream_t s = (hipStream_t)stream;
    if (dst_t == nullptr) {
        VIPANT_REQUIRE((R * C) % 4 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0, VIPANT_EALIGN,
                       "cast_bf16: need numel %% 4 == 0 and aligned pointers");
        const int64_t n4 = R * C / 4;
        hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, src, (bf16_t*)dst, n4);
    } else {
        hipLaunchKernelGGL(cast_transpose_kernel, dim3((unsigned)ceil_div(C, 64), (unsigned)ceil_div(R, 64)), dim3(256), 0,
                           s, src, (bf16_t*)dst, (bf16_t*)dst_t, (int)R, (int)C);
    }
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_cast_bf16_multi(const float* const* src, uint16_t* const* dst, uint16_t* const* dst_t,
                                          const int32_t* R, const int32_t* C, const int32_t* tile_start, int64_t ntensors,
                                          int64_t total_tiles, void* stream) {
    VIPANT_REQUIRE(ntensors > 0 && total_tiles > 0 && total_tiles < (1ll << 31), VIPANT_EBADSHAPE, "cast_bf16_multi: empty");
    CastMulti a{src, (bf16_t* const*)dst, (bf16_t* const*)dst_t, R, C, tile_start, (int)ntensors};
    hipLaunchKernelGGL(cast_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, a);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_residual_add(const void* x, const uint16_t* add, float* out, int64_t n, int32_t stream_flags, void* stream) {
    VIPANT_REQUIRE(n > 0 && n % 4 == 0, VIPANT_EBADSHAPE, "residual_add: numel must be a positive multiple of 4");
    if (stream_flags & VIPANT_STREAM_IN_F16)
        hipLaunchKernelGGL(residual_add_kernel<_Float16>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)x, (const bf16_t*)add, out, n / 4);
    else
        hipLaunchKernelGGL(residual_add_kernel<float>, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (const bf16_t*)add, out, n / 4);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_conv_weight_prep(const float* w, uint16_t* out, int64_t O, int64_t Cin, int64_t khw,
                                           int32_t mean_channels, void* stream) {
    VIPANT_REQUIRE(O > 0 && Cin > 0 && khw > 0, VIPANT_EBADSHAPE, "conv_weight_prep: empty");
    const int64_t total = mean_channels ? O * khw : O * Cin * khw;
    hipLaunchKernelGGL(conv_weight_prep_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (bf16_t*)out, O, (int)Cin, (int)khw, mean_channels);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_conv_weight_grad(const float* g, float* wgrad, int64_t O, int64_t Cin, int64_t khw,
                                           int32_t accumulate, void* stream) {
    VIPANT_REQUIRE(O > 0 && Cin > 0 && khw > 0, VIPANT_EBADSHAPE, "conv_weight_grad: empty");
    hipLaunchKernelGGL(conv_weight_grad_kernel, dim3(grid_for(O * Cin * khw, 256)), dim3(256), 0, (hipStream_t)stream, g,
                       wgrad, O, (int)Cin, (int)khw, accumulate);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_im2col(const float* x, uint16_t* out, int64_t b, int64_t C, int64_t T, int64_t F, int64_t ph,
                                 int64_t pw, int64_t sh, int64_t sw, void* stream) {
    VIPANT_REQUIRE(b > 0 && T >= ph && F >= pw && pw % 8 == 0, VIPANT_EBADSHAPE,
                   "im2col: need T>=ph, F>=pw, pw%%8==0 (T=%ld F=%ld ph=%ld pw=%ld)", (long)T, (long)F, (long)ph, (long)pw);
    const int nrow = (int)((T - ph) / sh + 1), ncol = (int)((F - pw) / sw + 1);
    const int vec_ok = (F % 4 == 0 && sw % 4 == 0 && (uintptr_t)x % 16 == 0) ? 1 : 0;
    const int64_t total = b * nrow * ncol * (C * ph * pw / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out,
                       (int)b, (int)C, (int)T, (int)F, (int)ph, (int)pw, (int)sh, (int)sw, nrow, ncol, vec_ok);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_assemble_tokens(const float* patches, const float* cls, const float* pos, float* tokens,
                                          int64_t b, int64_t P, int64_t D, void* stream) {
    VIPANT_REQUIRE(b > 0 && P > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "assemble_tokens: bad shape");
    hipLaunchKernelGGL(assemble_kernel, dim3(grid_for(b * (P + 1) * D / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       patches, cls, pos, tokens, b, (int)P, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_tokens_cls_rows(const float* cls, const float* pos, float* tokens, int64_t b, int64_t S, int64_t D,
                                          void* stream) {
    VIPANT_REQUIRE(b > 0 && S > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "tokens_cls_rows: bad shape");
    hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for(b * D / 4, 256)), dim3(256), 0, (hipStream_t)stream, cls, pos, tokens, b, S, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_assemble_tokens_bwd(const float* dtokens, uint16_t* dpatches, float* dcls, float* dpos,
                                              int32_t accumulate, int64_t b, int64_t P, int64_t D, void* stream) {
    VIPANT_REQUIRE(b > 0 && P > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "assemble_tokens_bwd: bad shape");
    hipLaunchKernelGGL(assemble_bwd_kernel, dim3(grid_for((P + 1) * D / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       dtokens, (bf16_t*)dpatches, dcls, dpos, accumulate, b, (int)P, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_l2norm_fwd(const float* x, float* y, float* norm, int64_t M, int64_t E, void* stream) {
    VIPANT_REQUIRE(M > 0 && E > 0 && E % 4 == 0, VIPANT_EBADSHAPE, "l2norm_fwd: bad shape");
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, x, y, norm, M, (int)E);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, uint16_t* dx_bf16,
                                     int64_t M, int64_t E, void* stream) {
    VIPANT_REQUIRE(M > 0 && E > 0 && E % 4 == 0, VIPANT_EBADSHAPE, "l2norm_bwd: bad shape");
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid_for(M, 4)), dim3(256), 0, (hipStream_t)stream, dy, y, norm, dx,
                       (bf16_t*)dx_bf16, M, (int)E);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_embed_tokens(const int64_t* tokens, const float* table, const float* pos, float* x, int64_t* eot,
                                       int64_t b, int64_t L, int64_t D, void* stream) {
    VIPANT_REQUIRE(b > 0 && L > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "embed_tokens: bad shape");
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for(b * L, 4)), dim3(256), 0, (hipStream_t)stream, tokens, table, pos,
                       x, eot, b, (int)L, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_gather_rows(const float* x, const int64_t* idx, float* out, int64_t n, int64_t rows_per_item,
                                      int64_t D, void* stream) {
    VIPANT_REQUIRE(n > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "gather_rows: bad shape");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, x, idx, out, n,
                       rows_per_item, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_scatter_rows(const float* g, const int64_t* idx, float* dx, int64_t n, int64_t rows_per_item,
                                       int64_t D, void* stream) {
    VIPANT_REQUIRE(n > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "scatter_rows: bad shape");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, g, idx, dx, n,
                       rows_per_item, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_scatter_rows_bf16(const float* g, const int64_t* idx, uint16_t* dx_bf16, uint16_t* compact_bf16, int64_t n,
                                            int64_t rows_per_item, int64_t D, void* stream) {
    VIPANT_REQUIRE(n > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "scatter_rows_bf16: bad shape");
    hipLaunchKernelGGL(scatter_rows_bf16_kernel, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, g, idx, (bf16_t*)dx_bf16,
                       (bf16_t*)compact_bf16, n, rows_per_item, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" size_t vipant_colsum_workspace_bytes(int64_t M, int64_t N) {
    (void)M;
    return (size_t)CS_ROWCHUNKS * (size_t)N * sizeof(float);
}

extern "C" int32_t vipant_colsum_bf16(const uint16_t* X, int64_t ldx, float* out, int64_t M, int64_t N, int32_t accumulate,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0, VIPANT_EBADSHAPE, "colsum: need N%%8==0, ldx%%8==0");
    VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= vipant_colsum_workspace_bytes(M, N), VIPANT_ENOWORKSPACE,
                   "colsum: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    // a row chunk per 32 rows, at most CS_ROWCHUNKS: `batch` rows (the read-out rows' gradients) are not spread over 128 partials
    int chunks = (int)(ceil_div(M, 32) < CS_ROWCHUNKS ? ceil_div(M, 32) : CS_ROWCHUNKS);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)ceil_div(N, 512), chunks), dim3(256), 0, s, (const bf16_t*)X, ldx,
                       (float*)workspace, M, (int)N);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, (const float*)workspace,
                       chunks, (int)N, out, accumulate);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}
