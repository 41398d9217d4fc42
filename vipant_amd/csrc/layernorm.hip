// LayerNorm forward / backward with fp32 statistics (clip/model.py:154-160, eps = 1e-5).
// HBM-bound: one wave per token row, 16-byte accesses, wave-shuffle reductions, fp32 in / bf16 out so
// that the normalised activations feed the next MFMA contraction without another cast pass.
#include <stdlib.h>
#include <type_traits>

#include "common.h"

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
namespace {

constexpr float LN_EPS = 1e-5f;

// The residual stream is fp32 or fp16 (the reference's own autocast precision, clip/model.py:157-160); statistics are fp32 always.
typedef _Float16 f16_t;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 load_stream4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 load_stream4(const f16_t* p) {
    const f16x4 h = *(const f16x4*)p;
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ void store_stream4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void store_stream4(f16_t* p, f32x4 v) {
    f16x4 h;
    h[0] = (f16_t)v[0]; h[1] = (f16_t)v[1]; h[2] = (f16_t)v[2]; h[3] = (f16_t)v[3];
    *(f16x4*)p = h;
}

// The block quantiser of vipant_quant_e4m3_mx (elementwise.hip; MX layout: common.h) on a row that is already in registers, four
// consecutive elements per lane and 256-column step -- a block of 32 is eight lanes: the values are first rounded to bf16, so bytes
// and scales are those the stand-alone kernel produces from the bf16 tensor this kernel also writes.
template <int NV>
__device__ __forceinline__ void quant_row_mx(f32x4 (&o)[NV], uint8_t* __restrict__ qrow, uint8_t* __restrict__ scales, int64_t row, int lane) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        const bf16x4 b = f32x4_to_bf16x4(o[t]);
        const u32x2 w = __builtin_bit_cast(u32x2, b);
        float sc;
        const int e = (int)mx_scale_byte<8>(mx_absmax2(mx_absmax2(0u, w[0]), w[1]), &sc) - 127;
        *(int*)(qrow + (t * 64 + lane) * 4) = mx_pack4_bf16(w[0], w[1], sc);
        if ((lane & 7) == 0) scales[mx_scale_offset(row, t * 8 + (lane >> 3), NV * 2)] = (uint8_t)(e + 127);
    }
}

// Q8: the e4m3 form of the output is written too.  A compile-time variant: the quantiser's registers in the plain kernel cost two waves
// per SIMD of occupancy (72 -> 85 VGPRs at D = 768) and 25 us of a 180-us launch that lives on memory latency (round 5).
template <int NV, typename XI, typename XO, bool Q8 = false>  // D = NV * 256; XI / XO: the stream's element type on the way in / out
__global__ __launch_bounds__(256) void ln_fwd_kernel(const XI* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ y32, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t M,
                                                     const bf16_t* __restrict__ add, XO* __restrict__ sum_out,
                                                     uint8_t* __restrict__ q8, uint8_t* __restrict__ q8s) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    f32x4 gv[NV], bv[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        gv[t] = *(const f32x4*)(gamma + (t * 64 + lane) * 4);
        bv[t] = *(const f32x4*)(beta + (t * 64 + lane) * 4);
    }
    // Q8 (round 6): STATIC scales, one per 32-column block and the same for every row -- block-uniform by construction, which is what the
    // e4m3 weight-gradient contraction needs of this matrix (vipant_gemm_tn_e4m3), and no reduction over the row at all.  A normalised
    // row has |xhat| <= sqrt(D - 1), so |y_c| <= sqrt(D) |gamma_c| + |beta_c|: nothing can saturate; the bound is about three binades
    // above what a typical row needs, which e4m3 pays for only below 2^-6 of the scale's unit (|xhat| < ~1e-3: the subnormal grid).
    const float sqrt_d = sqrtf((float)D);
    for (int64_t row = wave; row < M; row += nwaves) {
        const XI* xr = x + row * ldx;
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            v[t] = load_stream4(xr + (t * 64 + lane) * 4);
            if (add != nullptr) {      // residual add fused in front of the norm: v = x + branch (bf16)
                const bf16x4 a4 = *(const bf16x4*)(add + row * D + (t * 64 + lane) * 4);
                v[t] += f32x4{(float)a4[0], (float)a4[1], (float)a4[2], (float)a4[3]};
                // the new stream is stored in its own precision; the norm below works on the unrounded sum
                if (sum_out != nullptr) store_stream4(sum_out + row * D + (t * 64 + lane) * 4, v[t]);
            }
            s += v[t][0] + v[t][1] + v[t][2] + v[t][3];
        }
        const float mu = wave_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            v[t] -= mu;
            q += v[t][0] * v[t][0] + v[t][1] * v[t][1] + v[t][2] * v[t][2] + v[t][3] * v[t][3];
        }
        const float rs = rsqrtf(wave_sum(q) * (1.0f / D) + LN_EPS);
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const f32x4 o = v[t] * rs * gv[t] + bv[t];
            const int64_t off = row * D + (t * 64 + lane) * 4;
            if (y != nullptr) *(bf16x4*)(y + off) = f32x4_to_bf16x4(o);
            if (y32 != nullptr) *(f32x4*)(y32 + off) = o;
            v[t] = o;
        }
        if (Q8) {
            // (the static scales are re-derived from gamma / beta for every row -- a dozen VALU instructions per 32-column block in a
            // kernel that waits on memory -- rather than kept: four more live registers cost the D = 1024 kernel its fifth wave per
            // SIMD, 598 -> 665 us per launch)
            float sd = sqrt_d;
            asm volatile("" : "+v"(sd));         // (or the compiler hoists the whole derivation out of the row loop again)
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                float b = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) b = fmaxf(b, fabsf(gv[t][e]) * sd + fabsf(bv[t][e]));
                const uint32_t m = mx_lane_max_u<8>((__float_as_uint(b * 1.01f) + 0xFFFFu) >> 16);    // bf16 bits, rounded up; 8 lanes = 32 columns
                float sc;
                const uint32_t byte = mx_scale_of_max(m, &sc);
                const u32x2 w = __builtin_bit_cast(u32x2, f32x4_to_bf16x4(v[t]));
                *(int*)(q8 + row * D + (t * 64 + lane) * 4) = mx_pack4_bf16(w[0], w[1], sc);
                if ((lane & 7) == 0) q8s[mx_scale_offset(row, t * 8 + (lane >> 3), NV * 2)] = (uint8_t)byte;
            }
        }
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd.
// Per-block partial sums of dgamma = sum dy * xhat and dbeta = sum dy go to `partial` [grid, 2, D].
// DRES_BF16: the residual-stream gradient is a bf16 [M, D] tensor (read here, may be the same buffer as dxb: a lane reads its
// elements of a row before it writes them) instead of an fp32 one.
template <int NV, bool DY_F32, bool DRES_BF16, typename XT, int NWV, bool Q8 = false>
__global__ __launch_bounds__(NWV * 64) void ln_bwd_kernel(const void* __restrict__ dy_, const XT* __restrict__ x,
                                                     int64_t ldx, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const void* dres_, float* dx, int64_t lddx,
                                                     bf16_t* dxb, float* __restrict__ partial, int64_t M,
                                                     uint8_t* __restrict__ q8, uint8_t* __restrict__ q8s) {
    const float* dres = DRES_BF16 ? nullptr : (const float*)dres_;
    const bf16_t* dresb = DRES_BF16 ? (const bf16_t*)dres_ : nullptr;
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * NWV + wv;
    const int64_t nwaves = (int64_t)gridDim.x * NWV;
    f32x4 gv[NV], dg[NV], db[NV], dsum[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) {
        gv[t] = *(const f32x4*)(gamma + (t * 64 + lane) * 4);
        dg[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dsum[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Two rows in flight per wave: row r + nwaves is requested before row r is reduced.  With 512 persistent workgroups (8 waves
    // per CU) one row per wave left ~36 KB in flight per CU and the kernel ran at the latency, not at the bandwidth (4.75 TB/s).
    using DyRaw = typename std::conditional<DY_F32, f32x4, bf16x4>::type;
    using XRaw = typename std::conditional<sizeof(XT) == 4, f32x4, f16x4>::type;
    struct Raw { DyRaw dy[NV]; XRaw x[NV]; bf16x4 rb[NV]; f32x4 rf[NV]; float mu, rs; };
    auto load_row = [&](int64_t row, Raw& r) {
        r.mu = mean[row]; r.rs = rstd[row];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int col = (t * 64 + lane) * 4;
            r.dy[t] = *(const DyRaw*)((const typename std::conditional<DY_F32, float, bf16_t>::type*)dy_ + row * D + col);
            r.x[t] = *(const XRaw*)(x + row * ldx + col);
            if (DRES_BF16) r.rb[t] = *(const bf16x4*)(dresb + row * D + col);
            else if (dres != nullptr) r.rf[t] = *(const f32x4*)(dres + row * lddx + col);
        }
    };
    auto reduce_row = [&](int64_t row, const Raw& r) {
        const float mu = r.mu, rs = r.rs;
        f32x4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            f32x4 dyv, xv;
#pragma unroll
            for (int e = 0; e < 4; ++e) { dyv[e] = (float)r.dy[t][e]; xv[e] = (float)r.x[t][e]; }
            xh[t] = (xv - mu) * rs;
            g[t] = dyv * gv[t];
            dg[t] += dyv * xh[t];
            db[t] += dyv;
            s1 += g[t][0] + g[t][1] + g[t][2] + g[t][3];
            s2 += g[t][0] * xh[t][0] + g[t][1] * xh[t][1] + g[t][2] * xh[t][2] + g[t][3] * xh[t][3];
        }
        s1 = wave_sum(s1) * (1.0f / D);
        s2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
        for (int t = 0; t < NV; ++t) {
            const int col = (t * 64 + lane) * 4;
            f32x4 o = (g[t] - s1 - xh[t] * s2) * rs;
            if (DRES_BF16) o += f32x4{(float)r.rb[t][0], (float)r.rb[t][1], (float)r.rb[t][2], (float)r.rb[t][3]};
            else if (dres != nullptr) o += r.rf[t];
            dsum[t] += o;
            if (dx != nullptr) *(f32x4*)(dx + row * lddx + col) = o;
            if (dxb != nullptr) *(bf16x4*)(dxb + row * D + col) = f32x4_to_bf16x4(o);
            g[t] = o;
        }
        if (Q8) quant_row_mx<NV>(g, q8 + row * D, q8s, row, lane);
    };
    // a ring of RING rows per wave: row r + (RING - 1) nwaves is requested before row r is reduced (static indices: the loop
    // body is written out RING times)
    constexpr int RING = 2;          // measured: 4 rows in flight per wave are not faster (191.7 vs 190.3 us)
    Raw ring[RING];
#pragma unroll
    for (int i = 0; i < RING - 1; ++i)
        if (wave + i * nwaves < M) load_row(wave + i * nwaves, ring[i]);
    for (int64_t row = wave; row < M; row += RING * nwaves) {
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            const int64_t r = row + i * nwaves;
            if (r + (RING - 1) * nwaves < M) load_row(r + (RING - 1) * nwaves, ring[(i + RING - 1) % RING]);
            if (r < M) reduce_row(r, ring[i]);
        }
    }
    // block reduction of the 4 waves' column sums, one quantity at a time through a [4][D] LDS buffer (12 KiB at
    // D = 768: does not limit residency), then one row of partials per BLOCK
    __shared__ float red[NWV * D];
#pragma unroll
    for (int which = 0; which < 3; ++which) {
#pragma unroll
        for (int t = 0; t < NV; ++t)
            *(f32x4*)(red + wv * D + (t * 64 + lane) * 4) = which == 0 ? dg[t] : (which == 1 ? db[t] : dsum[t]);
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += NWV * 64) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < NWV; ++w) acc += red[w * D + i];
            partial[((int64_t)blockIdx.x * 3 + which) * D + i] = acc;
        }
        __syncthreads();
    }
}

// One workgroup per 64 columns: 16 row-lanes x 64 column-lanes sweep the per-block partials, LDS-reduce the lanes.
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* __restrict__ partial, int nblocks, int D,
                                                               float* dgamma, float* dbeta, float* dxcolsum,
                                                               int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;          // index into the [3 * D] (dgamma | dbeta | colsum dx) vector
    const int lim = dxcolsum != nullptr ? 3 * D : 2 * D;
    float s = 0.f;
    if (i < lim)
        for (int b = ty; b < nblocks; b += 16) s += partial[(int64_t)b * 3 * D + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < lim) {
        s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][tx];
        float* dst = i < D ? dgamma + i : (i < 2 * D ? dbeta + (i - D) : dxcolsum + (i - 2 * D));
        *dst = accumulate ? *dst + s : s;
    }
}

// persistent workgroups (the column sums accumulate in registers across a wave's rows), ONE per CU with two rows in flight per wave:
// at M = 161 792 with fp16 rows 188-190 us for 256 workgroups; 226 / 220 / 222 / 201-205 us for 192 / 320 / 384 / 512; 8-wave
// workgroups 200-206 us; the round-2 kernel (one row per wave, 512 workgroups) 226 us.  VIPANT_LN_BLOCKS overrides (timing only)
int ln_blocks(int64_t M) {
    static const int env = getenv("VIPANT_LN_BLOCKS") ? atoi(getenv("VIPANT_LN_BLOCKS")) : 256;      // timing experiments
    static const int cap = env < 1 ? 1 : env;          // (0 or a negative value would launch no workgroup)
    int64_t b = ceil_div(M, 16);
    return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" int32_t vipant_layernorm_fwd_e4m3(const void* x, int64_t ldx, const float* gamma, const float* beta,
                                             uint16_t* y, float* y_f32, float* mean, float* rstd, int64_t M, int64_t D,
                                             const uint16_t* add, void* sum_out, uint8_t* q, uint8_t* qscale, int32_t stream_flags,
                                             void* stream) {
    VIPANT_REQUIRE((q == nullptr) == (qscale == nullptr), VIPANT_EBADSHAPE, "layernorm_fwd: q and qscale go together");
    VIPANT_REQUIRE(M > 0 && D > 0 && D % 256 == 0 && D <= 1024, VIPANT_EBADSHAPE,
                   "layernorm_fwd: D must be a multiple of 256 up to 1024 (D=%ld)", (long)D);
    const bool in16 = (stream_flags & VIPANT_STREAM_IN_F16) != 0, out16 = (stream_flags & VIPANT_STREAM_OUT_F16) != 0;
    VIPANT_REQUIRE(ldx >= D && ldx % 4 == 0 && (uintptr_t)x % (in16 ? 8 : 16) == 0, VIPANT_EALIGN, "layernorm_fwd: bad ldx/alignment");
    hipStream_t s = (hipStream_t)stream;
    // one row per wave, no grid-stride loop: 254 us against 286 us with 2048 persistent workgroups at M = 161 792 (1.49 GB)
    const int blocks = (int)(ceil_div(M, 4) > (1 << 20) ? (1 << 20) : ceil_div(M, 4));
#define LN_FWD_Q(NV, XI, XO, Q)                                                                                                 \
    hipLaunchKernelGGL((ln_fwd_kernel<NV, XI, XO, Q>), dim3(blocks), dim3(256), 0, s, (const XI*)x, ldx, gamma, beta, (bf16_t*)y, \
                       y_f32, mean, rstd, M, (const bf16_t*)add, (XO*)sum_out, q, qscale)
    // (the quantising variants exist for the fp16 stream inside the stacks, which is where e4m3 towers run them)
#define LN_FWD_T(NV, XI, XO)                                                      \
    do {                                                                          \
        if (q != nullptr) LN_FWD_Q(NV, XI, XO, true); else LN_FWD_Q(NV, XI, XO, false); \
    } while (0)
#define LN_FWD(NV)                                                              \
    do {                                                                        \
        if (in16 && out16) LN_FWD_T(NV, f16_t, f16_t);                          \
        else if (in16) LN_FWD_T(NV, f16_t, float);                              \
        else if (out16) LN_FWD_T(NV, float, f16_t);                             \
        else LN_FWD_T(NV, float, float);                                        \
    } while (0)
    switch (D / 256) {
        case 1: LN_FWD(1); break;
        case 2: LN_FWD(2); break;
        case 3: LN_FWD(3); break;
        default: LN_FWD(4); break;
    }
#undef LN_FWD
#undef LN_FWD_T
#undef LN_FWD_Q
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta,
                                        uint16_t* y, float* y_f32, float* mean, float* rstd, int64_t M, int64_t D,
                                        const uint16_t* add, float* sum_out, void* stream) {
    return vipant_layernorm_fwd_e4m3(x, ldx, gamma, beta, y, y_f32, mean, rstd, M, D, add, sum_out, nullptr, nullptr, 0, stream);
}

extern "C" size_t vipant_layernorm_bwd_workspace_bytes(int64_t M, int64_t D) {
    return (size_t)ln_blocks(M) * 3 * (size_t)D * sizeof(float);
}

extern "C" int32_t vipant_layernorm_bwd_e4m3(const void* dy, int32_t flags, const void* x, int64_t ldx,
                                             const float* mean, const float* rstd, const float* gamma, const void* dres,
                                             float* dx_f32, int64_t lddx, uint16_t* dx_bf16, float* dgamma, float* dbeta,
                                             float* dx_colsum, int32_t accumulate, int64_t M, int64_t D, void* workspace,
                                             size_t workspace_bytes, uint8_t* q, uint8_t* qscale, void* stream) {
    VIPANT_REQUIRE((q == nullptr) == (qscale == nullptr), VIPANT_EBADSHAPE, "layernorm_bwd: q and qscale go together");
    VIPANT_REQUIRE(M > 0 && D > 0 && D % 256 == 0 && D <= 1024, VIPANT_EBADSHAPE,
                   "layernorm_bwd: D must be a multiple of 256 up to 1024 (D=%ld)", (long)D);
    VIPANT_REQUIRE(ldx >= D && ldx % 4 == 0 && lddx >= D && lddx % 4 == 0, VIPANT_EALIGN, "layernorm_bwd: bad strides");
    VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= vipant_layernorm_bwd_workspace_bytes(M, D),
                   VIPANT_ENOWORKSPACE, "layernorm_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = ln_blocks(M);
    float* partial = (float*)workspace;
    const bool dy_is_f32 = (flags & VIPANT_LN_DY_F32) != 0, dres_bf16 = (flags & VIPANT_LN_DRES_BF16) != 0;
    VIPANT_REQUIRE(!(dres_bf16 && (dres == nullptr || dy_is_f32)), VIPANT_EBADSHAPE,
                   "layernorm_bwd: a bf16 residual gradient needs dres and a bf16 dy");
    const bool x16 = (flags & VIPANT_LN_X_F16) != 0;
#define LN_BWD_Q(NV, A, B, XT, Q)                                                                                                 \
    hipLaunchKernelGGL((ln_bwd_kernel<NV, A, B, XT, 4, Q>), dim3(blocks), dim3(256), 0, s, dy, (const XT*)x, ldx, mean, rstd, gamma, \
                       dres, dx_f32, lddx, (bf16_t*)dx_bf16, partial, M, q, qscale)
#define LN_BWD_T(NV, A, B, XT)                                                            \
    do {                                                                                  \
        if (q != nullptr) LN_BWD_Q(NV, A, B, XT, true); else LN_BWD_Q(NV, A, B, XT, false); \
    } while (0)
#define LN_BWD(NV)                                                                                                   \
    do {                                                                                                             \
        if (dy_is_f32) { if (x16) LN_BWD_T(NV, true, false, f16_t); else LN_BWD_T(NV, true, false, float); }         \
        else if (dres_bf16) { if (x16) LN_BWD_T(NV, false, true, f16_t); else LN_BWD_T(NV, false, true, float); }    \
        else { if (x16) LN_BWD_T(NV, false, false, f16_t); else LN_BWD_T(NV, false, false, float); }                 \
    } while (0)
    switch (D / 256) {
        case 1: LN_BWD(1); break;
        case 2: LN_BWD(2); break;
        case 3: LN_BWD(3); break;
        default: LN_BWD(4); break;
    }
#undef LN_BWD
#undef LN_BWD_T
#undef LN_BWD_Q
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((unsigned)ceil_div(3 * D, 64)), dim3(1024), 0, s,
                       (const float*)partial, blocks, (int)D, dgamma, dbeta, dx_colsum, accumulate);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

extern "C" int32_t vipant_layernorm_bwd(const void* dy, int32_t flags, const void* x, int64_t ldx,
                                        const float* mean, const float* rstd, const float* gamma, const void* dres,
                                        float* dx_f32, int64_t lddx, uint16_t* dx_bf16, float* dgamma, float* dbeta,
                                        float* dx_colsum, int32_t accumulate, int64_t M, int64_t D, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    return vipant_layernorm_bwd_e4m3(dy, flags, x, ldx, mean, rstd, gamma, dres, dx_f32, lddx, dx_bf16, dgamma, dbeta, dx_colsum,
                                     accumulate, M, D, workspace, workspace_bytes, nullptr, nullptr, stream);
}
