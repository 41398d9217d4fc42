// The last block of a tower evaluated on its READ-OUT ROWS only.
//
// Both read-outs of the path take one token per item from the stack's output -- ViTPostEncoder the class token
// (cvap/module/val.py:288-289), GPTPostEncoder the end-of-text token (val.py:143-145) -- so of the LAST block's output only
// `batch` of its `batch * S` rows are ever read, and only those rows carry a gradient.  Everything of that block that is per-token
// (out_proj, ln_2, the MLP, the query projection) therefore needs `batch` rows, forward and backward; what needs all tokens is
// ln_1 and the key / value projection (the read-out row attends to every token).  The attention itself is one query per
// (item, head): the two kernels here.  Results are those of the full block on the read-out rows (dead rows eliminated, nothing
// approximated); the contractions around them are the library's ordinary ones on `batch` rows (vipant_amd/ops.py, BackboneFn).
//
// HBM-bound streaming work: one wave per (item, head), eight lanes per key row, so every wave load / store covers eight whole
// 128-byte rows of the head's K / V (or dK / dV) block.
#include "common.h"

namespace {

// the read-out row of item i, clamped into [0, rows): an index outside the item's rows (a caller bug: the end-of-text search
// cannot produce one) must not turn into an out-of-bounds access of qkv / dqkv / the stream
__device__ __forceinline__ int row_index(const int64_t* idx, int64_t i, int rows) {
    if (idx == nullptr) return 0;
    const int64_t v = idx[i];
    return (int)(v < 0 ? 0 : (v >= rows ? rows - 1 : v));
}

constexpr int RW = 4;                 // waves (= (item, head) pairs) per workgroup
constexpr float LOG2E = 1.4426950408889634f;

// Lane layout of the key phases: 8 lanes share a key row (lane & 7 = its 16-byte piece, 8 head dimensions), 8 keys per wave
// instruction -- a wave load covers 8 whole 128-byte rows.  A dot product over the 64 dimensions is 8 FMAs per lane and a
// 3-step butterfly over the 8 lanes of the key.
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
}
__device__ __forceinline__ float dot8(const float (&a)[8], const float (&b)[8]) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        s0 = __builtin_fmaf(a[e], b[e], s0);
        s1 = __builtin_fmaf(a[e + 1], b[e + 1], s1);
    }
    return s0 + s1;
}
__device__ __forceinline__ float sum8(float v) {        // over the 8 lanes of a key; every lane gets the sum
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// out_rows[i, h*64 ..] = softmax(q_i,h . K_i,h^T / 8 [keys <= row, if causal]) V_i,h;  probs[i, h, :] = that softmax (fp32)
__global__ __launch_bounds__(RW * 64) void mha_rows_fwd_kernel(const bf16_t* __restrict__ q_rows, const bf16_t* __restrict__ qkv,
                                                               const int64_t* __restrict__ idx, bf16_t* __restrict__ out_rows,
                                                               float* __restrict__ probs, int batch, int S, int H, int causal) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * RW + wave;
    const bool active = pair < batch * H;
    const int i = active ? pair / H : 0, h = active ? pair % H : 0;
    const int D = H * 64;
    const int64_t ld = 3 * (int64_t)D;
    const int nkeys = causal ? row_index(idx, i, S) + 1 : S;
    float* sc = sm + wave * S;
    const bf16_t* kbase = qkv + (int64_t)i * S * ld + D + h * 64;
    const int kl = lane >> 3, piece = (lane & 7) * 8;
    if (active) {
        float q[8];
        load8(q_rows + (int64_t)i * D + h * 64 + piece, q);
        float mx = -INFINITY;
        // four key groups (32 keys) per iteration, their loads issued together: the loop is bound by load latency otherwise
        for (int j0 = 0; j0 < nkeys; j0 += 32) {
            float kv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 8 + kl;
#pragma unroll
                for (int e = 0; e < 8; ++e) kv[u][e] = 0.f;
                if (j < nkeys) load8(kbase + j * ld + piece, kv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 8 + kl;
                const float d = sum8(dot8(kv[u], q)) * 0.125f;
                const float s = j < nkeys ? d : -INFINITY;
                if ((lane & 7) == 0 && j < S) sc[j] = s;
                mx = fmaxf(mx, s);
            }
        }
        mx = wave_max(mx);
        // (the scores of a pair are written by some lanes of this wave and read by others: one wave's LDS operations execute in
        // program order, the wave barrier only keeps the compiler from moving a read above the writes)
        __builtin_amdgcn_wave_barrier();
        float sum = 0.f;
        for (int j = lane; j < S; j += 64) {
            const float p = j < nkeys ? __builtin_amdgcn_exp2f((sc[j] - mx) * LOG2E) : 0.f;
            sc[j] = p;
            sum += p;
        }
        const float inv = 1.0f / wave_sum(sum);
        for (int j = lane; j < S; j += 64) {
            const float p = sc[j] * inv;
            sc[j] = p;
            probs[(int64_t)pair * S + j] = p;
        }
    }
    __syncthreads();
    if (active) {
        // o = sum_j p_j V_j: each lane accumulates its 8 dimensions over the keys j = kl (mod 8), then the 8 key groups are summed
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j0 = kl; j0 < nkeys; j0 += 32) {
            float vv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[u][e] = 0.f;
                if (j0 + u * 8 < nkeys) load8(kbase + D + (j0 + u * 8) * ld + piece, vv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float p = j0 + u * 8 < nkeys ? sc[j0 + u * 8] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(p, vv[u][e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] += __shfl_xor(acc[e], 8, 64);
            acc[e] += __shfl_xor(acc[e], 16, 64);
            acc[e] += __shfl_xor(acc[e], 32, 64);
        }
        if (kl == 0) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
            *(bf16x8*)(out_rows + (int64_t)i * D + h * 64 + piece) = o;
        }
    }
}

// Backward of the above for one query per (item, head): with dp_j = do . V_j and delta = sum_j p_j dp_j (= do . o),
//   dV_j = p_j do,  ds_j = p_j (dp_j - delta),  dK_j = ds_j q / 8,  dq = sum_j ds_j K_j / 8.
// dK / dV are written for EVERY key row of the item (zero rows behind a causal limit) into the K and V column blocks of
// dqkv [batch * S, 3 D]; its Q column block is not touched (the query gradient exists for the read-out rows only: dq_rows).
__global__ __launch_bounds__(RW * 64) void mha_rows_bwd_kernel(const bf16_t* __restrict__ q_rows, const bf16_t* __restrict__ qkv,
                                                               const int64_t* __restrict__ idx, const float* __restrict__ probs,
                                                               const bf16_t* __restrict__ dout_rows, bf16_t* __restrict__ dq_rows,
                                                               bf16_t* __restrict__ dqkv, int batch, int S, int H, int causal) {
    extern __shared__ float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * RW + wave;
    const bool active = pair < batch * H;
    const int i = active ? pair / H : 0, h = active ? pair % H : 0;
    const int D = H * 64;
    const int64_t ld = 3 * (int64_t)D;
    const int nkeys = causal ? row_index(idx, i, S) + 1 : S;
    float* sc = sm + wave * S;
    const int64_t base = (int64_t)i * S * ld + D + h * 64;
    const bf16_t* kbase = qkv + base;
    const int kl = lane >> 3, piece = (lane & 7) * 8;
    if (active) {
        float dov[8], qv[8];
        load8(dout_rows + (int64_t)i * D + h * 64 + piece, dov);
        load8(q_rows + (int64_t)i * D + h * 64 + piece, qv);
        const float* pr = probs + (int64_t)pair * S;
        // dp_j = do . V_j for every key, delta = sum_j p_j dp_j
        float dl = 0.f;
        for (int j0 = 0; j0 < nkeys; j0 += 32) {
            float vv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 8 + kl;
#pragma unroll
                for (int e = 0; e < 8; ++e) vv[u][e] = 0.f;
                if (j < nkeys) load8(kbase + D + j * ld + piece, vv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 8 + kl;
                const float dp = sum8(dot8(vv[u], dov));
                if ((lane & 7) == 0 && j < nkeys) {
                    sc[j] = dp;
                    dl = __builtin_fmaf(pr[j], dp, dl);
                }
            }
        }
        const float delta = wave_sum(dl);
        __builtin_amdgcn_wave_barrier();
        // dK_j, dV_j rows (8 keys per wave store instruction, whole 128-byte rows), ds_j kept for the dq pass; dq accumulated
        // on the way: lane holds its 8 dimensions of sum_{j = kl mod 8} ds_j K_j
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < S; j0 += 32) {
            float kk[4][8], pv[4], dsv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 8 + kl;
                const bool live = j < nkeys;
#pragma unroll
                for (int e = 0; e < 8; ++e) kk[u][e] = 0.f;
                if (live) load8(kbase + j * ld + piece, kk[u]);
                pv[u] = live ? pr[j] : 0.f;
                dsv[u] = live ? sc[j] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 8 + kl;
                if (j >= S) continue;
                const float p = pv[u];
                const float ds = p * (dsv[u] - delta) * 0.125f;
                bf16x8 tk, tv;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    tk[e] = (bf16_t)(ds * qv[e]);
                    tv[e] = (bf16_t)(p * dov[e]);
                }
                bf16_t* dk = dqkv + base + j * ld + piece;
                *(bf16x8*)dk = tk;
                *(bf16x8*)(dk + D) = tv;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(ds, kk[u][e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] += __shfl_xor(acc[e], 8, 64);
            acc[e] += __shfl_xor(acc[e], 16, 64);
            acc[e] += __shfl_xor(acc[e], 32, 64);
        }
        if (kl == 0) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)acc[e];
            *(bf16x8*)(dq_rows + (int64_t)i * D + h * 64 + piece) = o;
        }
    }
}

// dst row i <- src row i * rpi + idx[i] (idx == NULL: + 0), rows of `row_bytes` bytes (multiple of 16): one wave per row
__global__ __launch_bounds__(256) void gather_rows_kernel(const char* __restrict__ src, const int64_t* __restrict__ idx,
                                                          char* __restrict__ dst, int64_t n, int64_t rpi, int64_t row_bytes) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = i * rpi + row_index(idx, i, (int)rpi);
        for (int64_t c = (int64_t)lane * 16; c < row_bytes; c += 64 * 16)
            *(u32x4*)(dst + i * row_bytes + c) = *(const u32x4*)(src + r * row_bytes + c);
    }
}

// x row (i * rpi + idx[i]) <- bf16(x row + add row i);  add fp32 (ADD_F32) or bf16, compact [n, D]
template <bool ADD_F32>
__global__ __launch_bounds__(256) void add_rows_kernel(bf16_t* __restrict__ x, const int64_t* __restrict__ idx,
                                                       const void* __restrict__ add, int64_t n, int64_t rpi, int D) {
    const int lane = threadIdx.x & 63;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
        const int64_t r = i * rpi + row_index(idx, i, (int)rpi);
        for (int c = lane * 4; c < D; c += 256) {
            bf16x4* px = (bf16x4*)(x + r * D + c);
            const bf16x4 v = *px;
            f32x4 a;
            if (ADD_F32) {
                a = *(const f32x4*)((const float*)add + i * D + c);
            } else {
                const bf16x4 t = *(const bf16x4*)((const bf16_t*)add + i * D + c);
                a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
            }
            *px = f32x4_to_bf16x4(f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]} + a);
        }
    }
}

int32_t check_rows(const void* q_rows, const void* qkv, int64_t batch, int64_t S, int64_t H) {
    VIPANT_REQUIRE(q_rows != nullptr && qkv != nullptr, VIPANT_EBADSHAPE, "mha_rows: null operand");
    VIPANT_REQUIRE(batch > 0 && S > 0 && H > 0 && S <= 4096 && batch * H < (1ll << 31) - RW, VIPANT_EBADSHAPE,
                   "mha_rows: bad shape batch=%ld S=%ld H=%ld (S <= 4096)", (long)batch, (long)S, (long)H);
    VIPANT_REQUIRE((uintptr_t)q_rows % 16 == 0 && (uintptr_t)qkv % 16 == 0, VIPANT_EALIGN, "mha_rows: operands must be 16-byte aligned");
    return VIPANT_OK;
}

unsigned rows_grid(int64_t n) {
    const int64_t g = ceil_div(n, 4);
    return (unsigned)(g < 4096 ? g : 4096);
}

}  // namespace

extern "C" int32_t vipant_mha_rows_fwd(const uint16_t* q_rows, const uint16_t* qkv, const int64_t* idx, uint16_t* out_rows,
                                       float* probs, int64_t batch, int64_t S, int64_t H, int32_t causal, void* stream) {
    if (int32_t e = check_rows(q_rows, qkv, batch, S, H)) return e;
    VIPANT_REQUIRE(out_rows != nullptr && probs != nullptr, VIPANT_EBADSHAPE, "mha_rows_fwd: null output");
    hipLaunchKernelGGL(mha_rows_fwd_kernel, dim3((unsigned)ceil_div(batch * H, RW)), dim3(RW * 64), (size_t)RW * S * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)q_rows, (const bf16_t*)qkv, idx, (bf16_t*)out_rows, probs, (int)batch,
                       (int)S, (int)H, (int)causal);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_mha_rows_bwd(const uint16_t* q_rows, const uint16_t* qkv, const int64_t* idx, const float* probs,
                                       const uint16_t* dout_rows, uint16_t* dq_rows, uint16_t* dqkv, int64_t batch, int64_t S,
                                       int64_t H, int32_t causal, void* stream) {
    if (int32_t e = check_rows(q_rows, qkv, batch, S, H)) return e;
    VIPANT_REQUIRE(probs != nullptr && dout_rows != nullptr && dq_rows != nullptr && dqkv != nullptr, VIPANT_EBADSHAPE,
                   "mha_rows_bwd: null operand");
    VIPANT_REQUIRE((uintptr_t)dout_rows % 16 == 0 && (uintptr_t)dqkv % 16 == 0, VIPANT_EALIGN,
                   "mha_rows_bwd: operands must be 16-byte aligned");
    hipLaunchKernelGGL(mha_rows_bwd_kernel, dim3((unsigned)ceil_div(batch * H, RW)), dim3(RW * 64), (size_t)RW * S * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)q_rows, (const bf16_t*)qkv, idx, probs, (const bf16_t*)dout_rows,
                       (bf16_t*)dq_rows, (bf16_t*)dqkv, (int)batch, (int)S, (int)H, (int)causal);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_gather_rows_bytes(const void* src, const int64_t* idx, void* dst, int64_t n, int64_t rows_per_item,
                                      int64_t row_bytes, void* stream) {
    VIPANT_REQUIRE(n > 0 && rows_per_item > 0 && row_bytes > 0 && row_bytes % 16 == 0, VIPANT_EBADSHAPE,
                   "gather_rows_bytes: need n > 0 and rows of a multiple of 16 bytes (row_bytes=%ld)", (long)row_bytes);
    VIPANT_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0, VIPANT_EALIGN, "gather_rows_bytes: operands must be 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(n)), dim3(256), 0, (hipStream_t)stream, (const char*)src, idx, (char*)dst, n,
                       rows_per_item, row_bytes);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_add_rows_bf16(uint16_t* x, const int64_t* idx, const void* add, int32_t add_is_f32, int64_t n,
                                        int64_t rows_per_item, int64_t D, void* stream) {
    VIPANT_REQUIRE(n > 0 && rows_per_item > 0 && D > 0 && D % 4 == 0, VIPANT_EBADSHAPE, "add_rows_bf16: bad shape");
    if (add_is_f32)
        hipLaunchKernelGGL(add_rows_kernel<true>, dim3(rows_grid(n)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, idx, add, n,
                           rows_per_item, (int)D);
    else
        hipLaunchKernelGGL(add_rows_kernel<false>, dim3(rows_grid(n)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, idx, add, n,
                           rows_per_item, (int)D);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}
