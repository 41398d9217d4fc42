// C[M,N] = A[M,K] . B[N,K]^T with fused epilogues -- the dense contraction behind every nn.Linear /
// conv-as-GEMM on the path (cvap/module/val.py:245-247, 500-506; forward and the dX backward).
//
// gfx950 design: 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave),
// BK = 64, two 64 KiB LDS stages filled by LDS-DMA (buffer_load ... lds, 16 B per lane, out-of-range
// rows read as zero through the buffer descriptor), XOR-swizzled 128-B rows so that ds_read_b128
// fragment reads are bank-conflict free (swizzle applied on the DMA source address and on the read
// address: the LDS image of a DMA is lane-linear), v_mfma_f32_16x16x32_bf16 with the operands swapped
// (D = B_tile . A_tile^T) so that every lane ends up holding 4 consecutive output columns of one row.
#include <stdlib.h>

#include "common.h"
#include "nt_core.h"

namespace {

using namespace ntcore;

struct GemmNT {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; void* aux;
    int64_t lda, ldb, ldc;
    int M, N, K;
    float alpha;
    int dbg;   // VIPANT_GEMM_VARIANT (timing experiments only): bit 0 = skip the epilogue stores
    // fp8 operands only: E8M0 exponents (value = e4m3 * 2^(byte - 127)) -- A: one per 32 consecutive k of a row, in the MX layout of
    // common.h; B (a weight matrix): one per row
    const uint8_t* sa; const uint8_t* sb;
    // token assembly (EPI_F32 of the plain kernel only; ViTPreEncoder, cvap/module/val.py:249-257): with tok_p > 0, row m = (item,
    // patch) of the product lands in row item * (tok_p + 1) + patch + 1 of C and gets pos[patch + 1, :] added -- the patch
    // embedding written straight into the token matrix, whose class-token rows a one-row-per-item kernel fills
    int tok_p; const float* pos;
    // few-rows kernel only: blockIdx.y = h picks one of `nb` independent products; operand h starts h * stride elements further
    int64_t stride_a, stride_b, stride_c, stride_bias;
    // ping-pong kernel only: the stream's ticket block (common.h) -- tiles beyond a workgroup's first two are drawn from its XCD's
    // queue instead of the static stride; NULL: static walk
    uint32_t* tk = nullptr;
    uint32_t* tk_other = nullptr;      // the stream's other counter set: zeroed by this launch for the next one
    // few-rows kernel, vipant_gemm_nt_heads only: operands carried as bf16 PAIRS x = hi + lo (two planes, the lo plane `a_lo` / `c_lo`
    // elements behind the hi plane; 0: a single bf16 plane).  A pair in A doubles the K range (both planes against the same B rows);
    // a pair in C keeps 16 of the accumulator's 24 mantissa bits.
    int64_t a_lo = 0, c_lo = 0;
    // e4m3 kernels with EMIT: the epilogue also leaves the e4m3 form of its bf16 result [M, N] (bytes `cq`, MX block scales `cqs`) -- the A
    // operand of the next contraction, quantised where it is produced.  With cq set, C (and the code matrix `aux` of the QuickGELU
    // epilogue) may be NULL: only the e4m3 form is wanted (`running.recompute_mlp`: the forward keeps neither g nor the codes).
    uint8_t* cq = nullptr; uint8_t* cqs = nullptr;
};

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = tile / ntn, tn = tile % ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[8][4];
    mainloop(smem, p.A, p.lda, p.M, p.B, p.ldb, p.N, p.K, m0, n0, wave, lane, acc);

    // Epilogue: lane holds C[m][n4 .. n4+3] for m = m0 + wm*128 + i*16 + (lane&15),
    // n4 = n0 + wn*64 + j*16 + (lane>>4)*4.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n4 = n0 + wn * 64 + j * 16 + fq * 4;
        if (n4 >= p.N || (p.dbg & 1)) continue;
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (EPI != VIPANT_EPI_DQUICKGELU && EPI != VIPANT_EPI_SCALE_F32 && p.bias != nullptr)
            bv = *(const f32x4*)(p.bias + n4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm * 128 + i * 16 + frow;
            if (m >= p.M) continue;
            int64_t o = (int64_t)m * p.ldc + n4;
            f32x4 v = acc[i][j] + bv;
            if (EPI == VIPANT_EPI_F32 && p.tok_p > 0) {
                const int item = m / p.tok_p, patch = m - item * p.tok_p;
                o = ((int64_t)item * (p.tok_p + 1) + patch + 1) * p.ldc + n4;
                v += *(const f32x4*)(p.pos + (int64_t)(patch + 1) * p.N + n4);
            }
            if (EPI == VIPANT_EPI_BF16) {
                *(bf16x4*)((bf16_t*)p.C + o) = f32x4_to_bf16x4(v);
            } else if (EPI == VIPANT_EPI_F32) {
                *(f32x4*)((float*)p.C + o) = v;
            } else if (EPI == VIPANT_EPI_RESIDUAL_F32) {
                const f32x4 r = *(const f32x4*)((const float*)p.aux + o);
                *(f32x4*)((float*)p.C + o) = v + r;
            } else if (EPI == VIPANT_EPI_QUICKGELU) {
                *(bf16x4*)((bf16_t*)p.aux + o) = f32x4_to_bf16x4(v);
                f32x4 g;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = v[e] * quickgelu_gate(v[e]);
                *(bf16x4*)((bf16_t*)p.C + o) = f32x4_to_bf16x4(g);
            } else if (EPI == VIPANT_EPI_DQUICKGELU) {
                const bf16x4 u4 = *(const bf16x4*)((const bf16_t*)p.aux + o);
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (float)u4[e];
                    const float sg = quickgelu_gate(u);
                    d[e] = acc[i][j][e] * (sg * (1.0f + 1.702f * u * (1.0f - sg)));
                }
                *(bf16x4*)((bf16_t*)p.C + o) = f32x4_to_bf16x4(d);
            } else {  // VIPANT_EPI_SCALE_F32
                *(f32x4*)((float*)p.C + o) = acc[i][j] * p.alpha;
            }
        }
    }
}

// Stage the 256x256 accumulator tile through a 64 KiB LDS area and write it out as whole rows (16 B per lane),
// applying the epilogue on the way.  Uses raw s_barrier + lgkmcnt waits only, so LDS-DMA prefetches in flight survive.
template <int EPI>
__device__ __forceinline__ void load_bias(const GemmNT& p, int cn0, int wn, int fq, f32x4 (&bv)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n4 = cn0 + wn * 64 + j * 16 + fq * 4;
        bv[j] = (EPI != VIPANT_EPI_DQUICKGELU && EPI != VIPANT_EPI_DQUICKGELU_D8 && p.bias != nullptr && n4 < p.N) ? *(const f32x4*)(p.bias + n4)
                                                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <int EPI>
__device__ __forceinline__ void staged_epilogue(const GemmNT& p, f32x4 (&acc)[8][4], char* stg, int cm0, int cn0, int wm,
                                                int wn, int frow, int fq, int tid, const f32x4 (&bv)[4]) {
    if (EPI == VIPANT_EPI_RESIDUAL_F32) {
        // fp32 tile: 4 rounds of 64 rows x 1 KiB; 16-B chunk index XOR (row & 7)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (wm == (q >> 1)) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = (q & 1) * 4 + ii;
                    const int row = ii * 16 + frow;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ch = (wn * 16 + j * 4 + fq) ^ (row & 7);
                        *(f32x4*)(stg + row * 1024 + ch * 16) = acc[i][j] + bv[j];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int idx = t * 512 + tid;
                const int row = idx >> 6, ch = idx & 63;
                const int m = cm0 + q * 64 + row, n = cn0 + ch * 4;
                if (m < p.M && n < p.N) {
                    const f32x4 v = *(const f32x4*)(stg + row * 1024 + ((ch ^ (row & 7)) << 4));
                    const int64_t o = (int64_t)m * p.ldc + n;
                    *(f32x4*)((float*)p.C + o) = v + *(const f32x4*)((const float*)p.aux + o);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // bf16 tile: 2 rounds of 128 rows x 512 B; 16-B chunk index XOR (row & 7)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = i * 16 + frow;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = wn * 64 + j * 16 + fq * 4;
                        const int ch = (col >> 3) ^ (row & 7);
                        *(bf16x4*)(stg + row * 512 + ch * 16 + (col & 4) * 2) = f32x4_to_bf16x4(acc[i][j] + bv[j]);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int idx = t * 512 + tid;
                const int row = idx >> 5, ch = idx & 31;
                const int m = cm0 + h * 128 + row, n = cn0 + ch * 8;
                if (m < p.M && n < p.N) {
                    const bf16x8 v = *(const bf16x8*)(stg + row * 512 + ((ch ^ (row & 7)) << 4));
                    const int64_t o = (int64_t)m * p.ldc + n;
                    if (EPI == VIPANT_EPI_BF16) {
                        *(bf16x8*)((bf16_t*)p.C + o) = v;
                    } else if (EPI == VIPANT_EPI_QUICKGELU) {
                        *(bf16x8*)((bf16_t*)p.aux + o) = v;
                        bf16x8 g;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float u = (float)v[e];
                            g[e] = (bf16_t)(u * quickgelu_gate(u));
                        }
                        *(bf16x8*)((bf16_t*)p.C + o) = g;
                    } else {  // VIPANT_EPI_DQUICKGELU
                        const bf16x8 u8 = *(const bf16x8*)((const bf16_t*)p.aux + o);
                        bf16x8 d;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float u = (float)u8[e];
                            const float sg = quickgelu_gate(u);
                            d[e] = (bf16_t)((float)v[e] * (sg * (1.0f + 1.702f * u * (1.0f - sg))));
                        }
                        *(bf16x8*)((bf16_t*)p.C + o) = d;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Persistent variant for the large token-major contractions (M >> 256).  Measured on MI355X (tools/gemm_bench.py):
// the 2-stage main loop alone sustains ~1.1 PFLOP/s, but with K = 768 a tile's epilogue (8-byte partial-line
// stores straight from the accumulator layout) plus its un-overlapped prologue cost more than its 12 K-steps.
// Here each workgroup walks a list of tiles and
//   * stages the output tile through LDS (XOR-swizzled) so that global stores are whole 512-B / 1-KiB rows,
//     16 B per lane;  QuickGELU and its derivative are applied on the way out of LDS, so `u` is staged once;
//   * issues the LDS-DMA for the NEXT tile's first K-step before the epilogue, into the stage buffer the
//     epilogue does not use, so the prologue latency and the store drain overlap (raw s_barrier in the epilogue:
//     __syncthreads() would drain the DMA).
template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_persistent_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fq = lane >> 4, fs = (lane >> 1) & 7;

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    const int ntiles = ntm * ntn;
    const int G = gridDim.x;                                   // multiple of 8
    const int lane_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);   // XCD-contiguous position in a round

    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffA[i] = (uint32_t)(r * p.lda * 2 + c * 16);
        voffB[i] = (uint32_t)(r * p.ldb * 2 + c * 16);
    }
    uint32_t offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t cb = (uint32_t)(((ks * 4 + fq) ^ fs) << 4);
        offA[ks] = (uint32_t)((wm * 128 + frow) * 128) + cb;
        offB[ks] = (uint32_t)(A_BYTES + (wn * 64 + frow) * 128) + cb;
    }
    const int nk = p.K / BK;

    auto tile_rsrc = [&](int tile, __amdgpu_buffer_rsrc_t& rsA, __amdgpu_buffer_rsrc_t& rsB, int& m0, int& n0) {
        const int tm = tile / ntn, tn = tile % ntn;
        m0 = tm * BM; n0 = tn * BN;
        const int64_t a_bytes = ((int64_t)(p.M - m0) * p.lda - (p.lda - p.K)) * 2;
        const int64_t b_bytes = ((int64_t)(p.N - n0) * p.ldb - (p.ldb - p.K)) * 2;
        rsA = make_rsrc(p.A + (int64_t)m0 * p.lda, (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes));
        rsB = make_rsrc(p.B + (int64_t)n0 * p.ldb, (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes));
    };
    auto stage_load = [&](const __amdgpu_buffer_rsrc_t& rsA, const __amdgpu_buffer_rsrc_t& rsB, int stage, int kt) {
        char* sA = smem + stage * STAGE_BYTES + wave * 4096;
        char* sB = sA + A_BYTES;
        const uint32_t koff = (uint32_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, sA + i * 1024, voffA[i], koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsB, sB + i * 1024, voffB[i], koff);
    };

    int tile = lane_pos;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    int m0 = 0, n0 = 0;
    if (tile < ntiles) {
        tile_rsrc(tile, rsA, rsB, m0, n0);
        stage_load(rsA, rsB, 0, 0);
    }
    while (tile < ntiles) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                       // K-step 0 of this tile has landed in stage 0
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk && !(p.dbg & 4)) stage_load(rsA, rsB, (kt + 1) & 1, kt + 1);
            const char* s = smem + (kt & 1) * STAGE_BYTES;
            // Fragment pipeline: left alone, hipcc sinks every A-fragment read to just before its 4 MFMAs and waits
            // lgkmcnt(0) there (minimal registers, LDS latency exposed per MFMA group).  Keep the A reads two groups
            // ahead in a 3-deep register ring and pin the order with sched_barrier between groups.
            bf16x8 b[2][4], aq[3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[ks][j] = *(const bf16x8*)(s + offB[ks] + j * 2048);
            aq[0] = *(const bf16x8*)(s + offA[0]);
            aq[1] = *(const bf16x8*)(s + offA[0] + 2048);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ks = t >> 3, i = t & 7;
                if (t + 2 < 16) aq[(t + 2) % 3] = *(const bf16x8*)(s + offA[(t + 2) >> 3] + ((t + 2) & 7) * 2048);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][j], aq[t % 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(p.dbg & 8)) __syncthreads();
        }
        // both stages are free now: prefetch the next tile's first K-step into stage 0, stage the output through stage 1
        const int cm0 = m0, cn0 = n0;
        const int next = tile + G;
        // the bias is fetched BEFORE the next tile's LDS-DMA is issued: vmcnt retires in issue order, so a bias load issued
        // after the prefetch would make the epilogue wait for the prefetch to land before it can start
        f32x4 bv[4];
        load_bias<EPI>(p, cn0, wn, fq, bv);
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bv[j]));      // ... and awaited here, while nothing else is in flight
        if (next < ntiles) {
            tile_rsrc(next, rsA, rsB, m0, n0);
            stage_load(rsA, rsB, 0, 0);
        }
        staged_epilogue<EPI>(p, acc, smem + STAGE_BYTES, cm0, cn0, wm, wn, frow, fq, tid, bv);
        tile = next;
    }
}

// QuickGELU'(u) = sg (1 + 1.702 u (1 - sg)), sg = sigmoid(1.702 u), lies in (-0.0998, 1.0998).  The backward needs only this
// derivative, never u itself, so the forward can leave an 8-bit linear code of it instead of the bf16 pre-activation (1 instead of
// 2 bytes per element in both two-output epilogues): code = round((d + 0.1) * 212.5), step 4.7e-3, error <= 2.4e-3 -- the size of
// the bf16 rounding du carries anyway.
__device__ __forceinline__ uint32_t gelu_code(float d) {
    const float c = fminf(fmaxf((d + 0.1f) * 212.5f + 0.5f, 0.f), 255.f);
    return (uint32_t)c;
}
// the same, converted and inserted into byte `pos` of `word` by one v_cvt_pk_u8_f32 (round to nearest, saturating)
__device__ __forceinline__ uint32_t gelu_code_pack(float d, int pos, uint32_t word) {
    return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(d, 212.5f, 21.25f), (uint32_t)pos, word);
}
__device__ __forceinline__ float gelu_decode(uint32_t c) { return (float)c * (1.0f / 212.5f) - 0.1f; }
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------
// Ping-pong persistent kernel (bf16-output epilogues, K >= 128).
//
// In the kernel above both waves of a SIMD do the same thing at the same time: at the top of every K-tile all eight
// waves issue their LDS-DMA burst (8 x ~150 cycles of issue each) and no MFMA runs; measured, the loop without its DMA
// is 25-30 % faster.  Here the two wave groups of a workgroup (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; wave w and
// w+4 share a SIMD) run the SAME instruction stream HALF A K-TILE APART, separated by the workgroup barrier: while one
// group issues its DMA and starts the first 32-deep half of its K-tile, the other is in the MFMA-only second half, so
// every SIMD always has a wave that can feed the matrix pipe.
//
// What makes the lag legal with 160 KiB of LDS:
//   * A rows are private to a group (only waves with wm = g read rows g*128..+127): two 16 KiB stages per group, filled
//     by the group itself one K-tile ahead;
//   * B is read by both groups, so the slot of K-tile k is still being read by the lagging group when the leading one
//     wants K-tile k+1 filled: B lives in a ring of THREE 32 KiB slots.  Group 0 fills rows 0-127 of slot (k+1) at the
//     start of its K-tile k; group 1 fills rows 128-255 of slot (k+2) at the start of ITS K-tile k (half a K-tile later),
//     which gives every fill at least one full K-tile to land before its first reader and never touches a slot that any
//     wave can still be reading;
//   * the K-tile stream runs on across tile boundaries (next tile's descriptors, K offset 0), the epilogue sits between
//     two tiles' K-loops and stages through the one A stage of the group that the stream does not need (4 rounds of
//     32 rows), so both groups execute the same number of barriers per tile and stay exactly one barrier apart;
//   * every wave issues the same number of DMA instructions per K-tile whatever happens (past the last tile they go
//     through a zero-length descriptor: no traffic), so the counted vmcnt waits are always exact.
// Barrier intervals per wave and K-tile k: [DMA issue, 32 MFMAs (k-step 0)] | [32 MFMAs (k-step 1)].
// Waits before the barrier that ends an interval -- group 0: none | vmcnt(0);  group 1: vmcnt(8) | vmcnt(4).
constexpr int PP_A_STAGE = 128 * BK * 2;                 // 16 KiB
constexpr int PP_B_BASE = 4 * PP_A_STAGE;                // 64 KiB: [group][stage]
constexpr int PP_B_SLOT = BN * BK * 2;                   // 32 KiB
constexpr int PP_LDS_BYTES = PP_B_BASE + 3 * PP_B_SLOT;  // 160 KiB

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// The tile epilogue of the ping-pong kernels: 4 rounds of 32 rows of the group's 128 x 256 accumulator block staged through 16 KiB
// of LDS (`stg`, free for this group at this point of its stream) and written out row-contiguous, with the epilogue arithmetic.
// the 8-bit QuickGELU' codes one thread needs for epilogue round r of its group's block (four 8-byte loads)
__device__ __forceinline__ void load_codes(const GemmNT& p, int m0, int n0, int grp, int tl, int r, u32x2 (&cn)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int idx = t * 256 + tl;
        const int R = idx >> 5, ch = idx & 31;
        const int m = m0 + grp * 128 + (r * 2 + (R >> 4)) * 16 + (R & 15), n = n0 + ch * 8;
        cn[t] = (m < p.M && n < p.N) ? *(const u32x2*)((const uint8_t*)p.aux + (int64_t)m * p.ldc + n) : u32x2{0u, 0u};
    }
}

// `pre`: the codes of round 0, requested by the caller during the tile's last K-tile (QuickGELU' launch on the DEEP schedule: the
// HBM round trip of the first round's codes is then off the epilogue's critical path), or NULL
// EMIT: 0 = the bf16 result only; 1 = also its e4m3 form (p.cq, p.cqs); 2 = the e4m3 form ALONE (QuickGELU epilogue of a tower that
// keeps neither g nor the derivative codes: their arithmetic is not compiled in); 3 = the e4m3 form and the derivative codes, no bf16
// result (round 6: the e4m3 form is all the backward reads of g -- the weight-gradient contraction takes it as it is)
template <int EPI, int EMIT = 0>
__device__ __forceinline__ void pp_epilogue(const GemmNT& p, f32x4 (&acc)[8][4], const f32x4 (&bv)[4], char* stg, char* stg_hi,
                                            int m0, int n0, int grp, int wl, int frow, int fq, int tid,
                                            const u32x2 (*pre)[4] = nullptr, uint32_t* xch = nullptr) {
    // EMIT (round 6): the e4m3 form carries BLOCK-UNIFORM scales -- one per aligned block of 32 rows x 32 columns, the operand format of
    // the weight-gradient contraction (vipant_gemm_tn_e4m3), and as good a scale for each of the block's rows in the NT contraction that
    // reads this matrix next.  The scale comes from a BOUND the wave has in registers before anything is staged: a block is row tiles
    // 2 r, 2 r + 1 x column tiles 2 jp, 2 jp + 1 of one wave, |QuickGELU(u)| <= |u| and |QuickGELU'| <= 1.1, so max |acc + bias|
    // (x 1.1) over the wave bounds the block's results (at most a binade above the exact maximum: e4m3's relative precision does not
    // care, its 15 binades of normal range barely).  One wave reduction per block, all eight of a tile up front; the bytes travel to
    // the threads that quantise -- another mapping, other waves -- through 32 words per group of the B slot the last K-tile has just
    // vacated (`xch`), published by the staging barrier of round 0: no barrier of their own.
    if (EMIT) {
        constexpr bool DGELU = EPI == VIPANT_EPI_DQUICKGELU || EPI == VIPANT_EPI_DQUICKGELU_D8;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                float b = 0.f;
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const f32x4 u = acc[2 * r + ii][2 * jp + jj] + bv[2 * jp + jj];
                        b = fmaxf(fmaxf(b, fmaxf(fabsf(u[0]), fabsf(u[1]))), fmaxf(fabsf(u[2]), fabsf(u[3])));
                    }
                // (DPP + readlane reduction: the result is a scalar, and so is everything derived from it)
                const float bw = __uint_as_float(wave_max_u(__float_as_uint(b))) * (DGELU ? 1.11f : 1.01f);    // (+ 1 %: the two roundings to bf16 on the way)
                float unused;
                const uint32_t byte = mx_scale_of_max((__float_as_uint(bw) + 0xFFFFu) >> 16, &unused);     // bf16 bits, rounded up
                if ((tid & 63) == 0) xch[r * 8 + wl * 2 + jp] = byte;
            }
    }
    // 8 consecutive bf16 results of row m from column n on (n % 8 == 0), `blk` = their 32-column block inside the tile's 256 columns
    auto emit8 = [&](const bf16x8& val, int m, int n, int ch, int r) {
        const u32x4 w = __builtin_bit_cast(u32x4, val);
        const uint32_t byte = xch[r * 8 + (ch >> 2)];
        const float sc = byte ? __uint_as_float(byte << 23) : 1.0f;
        *(int2*)(p.cq + (int64_t)m * p.N + n) = int2{mx_pack4_bf16(w[0], w[1], sc), mx_pack4_bf16(w[2], w[3], sc)};
        if ((ch & 3) == 0) p.cqs[mx_scale_offset(m, n >> 5, p.N >> 7)] = (uint8_t)byte;
    };
    // rows 0-15 of a round are staged at `stg`, rows 16-31 at `stg_hi` (the ring kernel has two free 8-KiB pieces, not one of 16)
    stg_hi -= 16 * 512;
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const int tl = tid & 255;
    constexpr bool GELU_OUT = EPI == VIPANT_EPI_QUICKGELU || EPI == VIPANT_EPI_QUICKGELU_D8;
    constexpr bool GELU_IN = EPI == VIPANT_EPI_DQUICKGELU || EPI == VIPANT_EPI_DQUICKGELU_D8;
    constexpr bool D8 = EPI == VIPANT_EPI_QUICKGELU_D8 || EPI == VIPANT_EPI_DQUICKGELU_D8;
    // QuickGELU' inputs of the next round (bf16 pre-activations, or their 8-bit derivative codes), loaded one round ahead
    bf16x8 un[4];
    u32x2 cn[4];
    auto load_aux = [&](int r) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = t * 256 + tl;
            const int R = idx >> 5, ch = idx & 31;
            const int m = m0 + grp * 128 + (r * 2 + (R >> 4)) * 16 + (R & 15), n = n0 + ch * 8;
            const bool ok = m < p.M && n < p.N;
            if (D8) cn[t] = ok ? *(const u32x2*)((const uint8_t*)p.aux + (int64_t)m * p.ldc + n) : u32x2{0u, 0u};
            else un[t] = ok ? *(const bf16x8*)((const bf16_t*)p.aux + (int64_t)m * p.ldc + n) : bf16x8{};
        }
    };
    if (GELU_IN) {
        if (pre != nullptr) {
#pragma unroll
            for (int t = 0; t < 4; ++t) cn[t] = (*pre)[t];
        } else {
            load_aux(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = r * 2 + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int colbyte = wl * 128 + j * 32 + fq * 8;
                *(bf16x4*)((ii ? stg_hi : stg) + (ii * 16 + frow) * 512 + (((colbyte >> 4) ^ frow) << 4) + (colbyte & 8)) =
                    f32x4_to_bf16x4(acc[i][j] + bv[j]);
            }
        }
        sync();
        bf16x8 u8[4];
        u32x2 c8[4];
        if (GELU_IN) {      // this round's inputs arrived during the previous round
#pragma unroll
            for (int t = 0; t < 4; ++t) { u8[t] = un[t]; c8[t] = cn[t]; }
            if (r + 1 < 4) load_aux(r + 1);
        }
        if (EPI == VIPANT_EPI_QUICKGELU_D8 && (p.N & 15) == 0 && (p.ldc & 15) == 0) {
            // a thread takes 16 consecutive columns, so that the code leaves as one 16-B store per thread (the store path
            // is bound by instructions as much as by bytes: 6 instead of 8 store instructions per thread and round)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int idx = t * 256 + tl;
                const int R = idx >> 4, cp = idx & 15;
                const int m = m0 + grp * 128 + (r * 2 + (R >> 4)) * 16 + (R & 15), n = n0 + cp * 16;
                if (m < p.M && n < p.N && !(p.dbg & 1)) {
                    const int64_t o = (int64_t)m * p.ldc + n;
                    uint32_t cw[4];
                    bf16x8 g2[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const bf16x8 v = *(const bf16x8*)((R & 16 ? stg_hi : stg) + R * 512 + (((cp * 2 + h) ^ (R & 15)) << 4));
                        bf16x8& g = g2[h];
                        cw[h * 2] = cw[h * 2 + 1] = 0u;
                        // two elements per instruction (v_pk_mul / v_pk_add / v_pk_fma): this epilogue is bound by its own
                        // arithmetic as much as by its stores.  code = 212.5 (sg + 1.702 ge (1 - sg)) + 21.25, constants folded.
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f32x2 u = f32x2{(float)v[e], (float)v[e + 1]};
                            const f32x2 a = u * -2.4554669595930157f;
                            const f32x2 b = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.0f;
                            const f32x2 sg = f32x2{__builtin_amdgcn_rcpf(b[0]), __builtin_amdgcn_rcpf(b[1])};
                            const f32x2 ge = u * sg;
                            g[e] = (bf16_t)ge[0];
                            g[e + 1] = (bf16_t)ge[1];
                            if (EMIT != 2) {
                                const f32x2 c = (ge * 361.675f) * (1.0f - sg) + (sg * 212.5f + 21.25f);
                                uint32_t& w = cw[h * 2 + (e >> 2)];
                                w = __builtin_amdgcn_cvt_pk_u8_f32(c[0], (uint32_t)(e & 3), w);
                                w = __builtin_amdgcn_cvt_pk_u8_f32(c[1], (uint32_t)((e + 1) & 3), w);
                            }
                        }
                        if (EMIT < 2) *(bf16x8*)((bf16_t*)p.C + o + h * 8) = g;
                    }
                    if (EMIT != 2) *(u32x4*)((uint8_t*)p.aux + o) = u32x4{cw[0], cw[1], cw[2], cw[3]};
                    if (EMIT) {         // 16 columns per thread: two threads per 32-column block
                        const u32x4 w0 = __builtin_bit_cast(u32x4, g2[0]), w1 = __builtin_bit_cast(u32x4, g2[1]);
                        const uint32_t byte = xch[r * 8 + (cp >> 1)];
                        const float sc = byte ? __uint_as_float(byte << 23) : 1.0f;
                        *(i32x4*)(p.cq + (int64_t)m * p.N + n) = i32x4{mx_pack4_bf16(w0[0], w0[1], sc), mx_pack4_bf16(w0[2], w0[3], sc),
                                                                       mx_pack4_bf16(w1[0], w1[1], sc), mx_pack4_bf16(w1[2], w1[3], sc)};
                        if ((cp & 1) == 0) p.cqs[mx_scale_offset(m, n >> 5, p.N >> 7)] = (uint8_t)byte;
                    }
                }
            }
            sync();
            continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = t * 256 + tl;
            const int R = idx >> 5, ch = idx & 31;
            const int m = m0 + grp * 128 + (r * 2 + (R >> 4)) * 16 + (R & 15), n = n0 + ch * 8;
            if (m < p.M && n < p.N && !(p.dbg & 1)) {
                const bf16x8 v = *(const bf16x8*)((R & 16 ? stg_hi : stg) + R * 512 + ((ch ^ (R & 15)) << 4));
                const int64_t o = (int64_t)m * p.ldc + n;
                if (EPI == VIPANT_EPI_BF16) {
                    *(bf16x8*)((bf16_t*)p.C + o) = v;
                    if (EMIT) emit8(v, m, n, ch, r);
                } else if (GELU_OUT) {
                    bf16x8 g;
                    uint32_t code[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float u = (float)v[e];
                        const float sg = quickgelu_gate(u);
                        g[e] = (bf16_t)(u * sg);
                        code[e] = gelu_code(sg * (1.0f + 1.702f * u * (1.0f - sg)));
                    }
                    if (EMIT == 2) {
                    } else if (D8)
                        *(u32x2*)((uint8_t*)p.aux + o) = u32x2{code[0] | code[1] << 8 | code[2] << 16 | code[3] << 24,
                                                               code[4] | code[5] << 8 | code[6] << 16 | code[7] << 24};
                    else
                        *(bf16x8*)((bf16_t*)p.aux + o) = v;
                    if (EMIT < 2) *(bf16x8*)((bf16_t*)p.C + o) = g;
                    if (EMIT) emit8(g, m, n, ch, r);
                } else {  // QuickGELU'
                    bf16x8 d;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float dg;
                        if (D8) {
                            dg = gelu_decode((c8[t][e >> 2] >> ((e & 3) * 8)) & 255u);
                        } else {
                            const float u = (float)u8[t][e];
                            const float sg = quickgelu_gate(u);
                            dg = sg * (1.0f + 1.702f * u * (1.0f - sg));
                        }
                        d[e] = (bf16_t)((float)v[e] * dg);
                    }
                    if (EMIT < 2) *(bf16x8*)((bf16_t*)p.C + o) = d;      // (EMIT 2: the e4m3 form alone -- the weight-gradient contraction and c_fc^T read that)
                    if (EMIT) emit8(d, m, n, ch, r);
                }
            }
        }
        sync();
    }
}

// v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 operands.  Operand layout (tools/probes/mx_layout_probe.hip, mx_scale_probe.hip,
// exact integer data on the MI355X): lane (r = lane & 15, q = lane >> 4) holds bytes k = 16 q .. 16 q + 15 of row r in dwords
// 0-3 and k = 64 + 16 q .. in dwords 4-7 -- the two 16-byte fragments the bf16 loop reads for its k-steps 0 and 1 of a 128-byte
// row -- and its scale byte applies to row r, k in [32 q, 32 q + 32).  The byte of the scale register is picked by an
// instruction immediate, so the call sites are spelled out with literal selectors (a `switch` on the unrolled loop index
// compiled, but left the MFMAs in blocks of their own, and the machine sinker then moved all 32 of a K-tile below both barriers).
template <int V> struct Int { static constexpr int value = V; };
#define VIPANT_BF(ACC, B4, A4) \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, i32x4(B4)), __builtin_bit_cast(bf16x8, i32x4(A4)), ACC, 0, 0, 0)
#define VIPANT_MX(ACC, A, B, OA, SA, OB, SB) \
    ACC = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, ACC, 0, 0, OA, (int)(SA), OB, (int)(SB))

// ES = bytes per operand element: 2 = bf16 (K-tile of 64), 1 = e4m3 with per-row power-of-two scales (K-tile of 128: the same
// 128-byte rows, the same LDS images, DMA stream and barrier schedule; half the MFMA instructions, each twice as long, for
// twice the K -- twice the FLOP per byte moved and per cycle).
// DYN: the ticket walk (compile-time: the static walk's kernels are byte for byte what they were before it existed).
template <int EPI, int VAR, int ES = 2, int EMIT = 0, bool DYN = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_pp_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wl = wave & 3;            // grp = wm, wl = wn
    const int frow = lane & 15, fq = lane >> 4, fs = (lane >> 1) & 7;

    const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
    // experiment (VAR 7 / 8): column-grouped walk -- the XCDs split into two column groups (each keeps HALF of the weight matrix,
    // meant to stay in its 4 MiB L2) times four row quarters; a workgroup's sequence number then maps to (panel, column) inside
    // its XCD's share.  Needs the full grid of 256 and an even number of column tiles.
    constexpr bool GROUPED = VAR == 8 || VAR == 12;
    const int G = gridDim.x;                                   // multiple of 8, <= ntiles rounded up
    const int cg = ntn / 2, ppx = (ntm + 3) / 4;
    const int ntiles = GROUPED ? ((ppx * cg + 31) / 32) * 256 : ntm * ntn;
    const int lane_pos = (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (lane_pos >= ntiles) return;                            // whole workgroup: no barrier has been executed yet
    const int nk = p.K / (ES == 2 ? BK : 2 * BK);
    // Ticket walk (common.h; the host sets p.tk only for a full grid whose every queue is longer than four rounds, K >= 256).  A
    // workgroup's first four tiles are the static walk's (positions b >> 3, + 32, + 64, + 96 of queue b & 7: nothing to wait for at
    // start-up); from the fifth on, positions come from the queue's counter.  The stream needs a tile's successor one tile ahead, the
    // LDS is full and `vmcnt` retires in order (anything slow in front of the counted LDS-DMA waits stalls them), so a ticket travels
    // without a wait of its own, through the three places of a tile where one is free:
    //   1. thread 0 draws it at the START of an epilogue (the atomic has the whole epilogue to return);
    //   2. in the next tile's bias round trip (the one full wait a tile has anyway: the value is there) it becomes a tile index, and
    //      thread 0 stores that into the workgroup's mailbox (two words in global memory, alternating) at the start of THAT
    //      tile's epilogue, in front of the tile's own stores;
    //   3. one epilogue later every wave requests the word BEHIND its tile's stores (asm load: the counted waits of the next K-loop
    //      retire those stores and the load with them) and reads the register in that tile's bias round trip: a scalar from there,
    //      `tile_nn`, the tile after the next, described one epilogue later.
    // A workgroup therefore holds claims on four tiles beyond the one it computes.  (First form, same round: the word was read with
    // an ordinary load inside the K-loop, two K-tiles before the tile's end -- a load there has one barrier interval to return,
    // and the loop got a second peeled K-tile: +3 ... +17 us per launch, gone with this form except on QuickGELU', see the draw.)
    // (gemm_nt.hip is compiled with the atomic optimizer off: it turns a uniform atomic into "first lane adds, wait, broadcast" -- a
    // full wait where the draw is issued.)
    // All of the walk's state lives in VECTOR registers on purpose (the `"+v"` launderings below): the K-loop keeps ~100 scalars busy
    // with buffer descriptors and offsets, and a dozen more made hipcc spill 27 of them into VGPR lanes -- 70 v_readlane / v_writelane
    // and 290 hazard s_nops in the loop: +5 % on the c_fc launch, whichever walk ran (round 5, same-box A/B against the round-4 tree).
    constexpr bool dyn = DYN;
    constexpr int NO_TILE = 0x3FFFFFFF;
    int xq = blockIdx.x & 7;
    int qlen_own = 0;
    uint64_t mbox = 0;                          // this workgroup's two mailbox words (global address)
    uint64_t tkq = 0;                           // its queue's counter
    if (DYN) {
        if (GROUPED) {
            int rows = ntm - (xq & 3) * ppx;
            rows = rows < 0 ? 0 : (rows > ppx ? ppx : rows);
            qlen_own = rows * cg;
        } else {
            int rem = (ntiles & 255) - xq * 32;
            rem = rem < 0 ? 0 : (rem > 32 ? 32 : rem);
            qlen_own = (ntiles >> 8) * 32 + rem;
        }
        mbox = (uint64_t)(p.tk + VIPANT_TICKET_MBOX + 2 * blockIdx.x);
        tkq = (uint64_t)(p.tk + xq);
        asm volatile("" : "+v"(xq), "+v"(qlen_own), "+v"(mbox), "+v"(tkq));
    }
    uint32_t tk_pend = 0u;                      // thread 0: the counter value of the last draw (positions 128 + value and 128 + value + 1)
    // the two launches with 30 tiles per workgroup (c_fc + QuickGELU, QuickGELU') take TWO CONSECUTIVE positions per draw: QuickGELU'
    // +18 -> +7 ... +13 us against its static twin, c_fc 10-18 us FASTER than its static twin (three runs, two boxes).  It is not the
    // halved number of atomics that pays: pairing positions p and p + 32 (the same workgroup's tiles of two rounds of the static walk,
    // still one atomic per two tiles) gives the single-draw times back.  Consecutive positions are neighbouring column tiles of ONE
    // row tile; computed one after the other by one workgroup instead of side by side by two, the A rows are fetched twice (+0.2 GB
    // per launch in the PMC counters) but the 32 workgroups of a queue spread over twice as many A row tiles at any moment --
    // fewer workgroups pulling the same lines out of the L2 at the same time.  (Walking the queue in column PHASES -- all rows for three, two or one of the six columns, then the
    // next -- spreads the workgroups the same way with single draws and loses 25-60 us: the A rows come back from HBM a phase later, not from the
    // caches 27 us later.)  Runs of three consecutive positions are slower again (+10 ... +18 us), runs of six (a workgroup
    // takes a whole row of column tiles) by 90-115 us.  The short-queue launches (7 to 22 tiles per workgroup) lose 8-12 us with pairs -- a
    // coarser claim at the end of a short queue -- and keep single draws.
    constexpr bool DRAW2 = EPI == VIPANT_EPI_QUICKGELU_D8 || EPI == VIPANT_EPI_DQUICKGELU_D8;
    int tk_half = 0;                            // thread 0 (DRAW2): the draw's second position is still to be used
    // plain launches with >= 8 column tiles per row (qkv: nine) draw pairs too, but single positions for the last 96 of a queue -- 22
    // tiles per workgroup are few enough for the coarser claim to show at the end (pairs throughout: +12 us against the static
    // twin; with the taper: 567 -> 551 us, 8 us FASTER than the static twin; bit 29 of VIPANT_GEMM_VARIANT: single draws)
    int tk_n = DRAW2 ? 2 : ((EPI == VIPANT_EPI_BF16 && !(p.dbg & (1 << 29)) && ntn >= 8) ? 2 : 1);
    const bool tk_taper = !DRAW2;
    int tk_dry = 0;                             // thread 0: the queue is empty, stop drawing
    int tk_par = 0;                             // the mailbox word this tile's bias round trip reads (it writes the other one)
    int tk_first = 1;                           // the first tile's round trip has nothing to read: its "ticket" is the third static tile
    if (DYN) asm volatile("" : "+v"(tk_dry), "+v"(tk_par), "+v"(tk_first), "+v"(tk_half), "+v"(tk_n));
    auto tk_tile = [&](uint32_t drawn) {
        const int pos = 128 + (int)drawn;
        if (pos < qlen_own) return tickets::tile_of(xq, pos);
        tk_dry = 1;
        return NO_TILE;
    };
    if (dyn && tid == 0) {
        tk_pend = tickets::take_g(tkq, (uint32_t)tk_n);       // this workgroup's fifth (and sixth) tile; not awaited before the first tile's bias round trip
        // the stream's OTHER counter set is at rest (its last user, the stream's previous ticket launch, is complete; the next one
        // starts after this launch): leave it zeroed for that launch -- nobody has to find out who finishes last
        if (blockIdx.x < 8) tickets::put(p.tk_other + blockIdx.x, 0u);
    }

    // DMA: wave fills row blocks wave*4 .. wave*4+3 (8 rows x 128 B) of the tile's A and B rows, as in the kernel above
    uint32_t voffA[2], voffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffA[i] = (uint32_t)(r * p.lda * ES + c * 16);
        voffB[i] = (uint32_t)(r * p.ldb * ES + c * 16);
    }
    const uint32_t waveA = (uint32_t)(wave * 32 * p.lda * ES), waveB = (uint32_t)(wave * 32 * p.ldb * ES);
    const uint32_t pairA = (uint32_t)(16 * p.lda * ES), pairB = (uint32_t)(16 * p.ldb * ES);
    char* const ldsA = smem + grp * (2 * PP_A_STAGE) + wl * 4096;          // + stage * PP_A_STAGE
    char* const ldsB = smem + PP_B_BASE + wave * 4096;                      // + slot * PP_B_SLOT
    // fragment reads
    uint32_t offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t cb = (uint32_t)(((ks * 4 + fq) ^ fs) << 4);
        offA[ks] = (uint32_t)(grp * (2 * PP_A_STAGE) + frow * 128) + cb;
        offB[ks] = (uint32_t)(PP_B_BASE + (wl * 64 + frow) * 128) + cb;
    }

    // a tile's operand panels as (base pointer, byte range): scalars, so that "this tile or the next one" is a scalar select and
    // the buffer descriptor built from it stays in SGPRs (a select between two descriptors made hipcc keep them in VGPRs and
    // wrap every LDS-DMA in a readfirstlane loop)
    struct TileDesc { const char* a; const char* b; uint32_t abytes, bbytes; int m0, n0; };
    auto describe = [&](int tile) {
        TileDesc d;
        int tm = tile / ntn, tn = tile % ntn;
        bool valid = tile < ntiles;
        // ... and the same on the e4m3 kernels' plain static walk (N = 1024 launches of the ViT-L tower: K = 4096 1196 -> 1169 us, K = 3072
        // 907 -> 884, K = 1024 391 -> 387; bit 28: off).  On the bf16 kernels' static twin it helps where a row has many column tiles
        // (qkv, nine: 554 -> 540 us) and costs 1-3 % where it has three (N = 768) -- the ticket walk takes it as paired draws, below.
        if (!GROUPED && ES == 1 && !(p.dbg & (1 << 28)) && valid) {
            const int q = (tile & 255) >> 5, pos = (tile & 31) + 32 * (tile >> 8);
            const int w = pos & 31, t = pos >> 5, rounds = ntiles >> 8;       // full rounds only
            if ((t | 1) < rounds) {
                const int t2 = tickets::tile_of(q, 64 * (t >> 1) + 2 * w + (t & 1));
                tm = t2 / ntn; tn = t2 % ntn;
            }
        }
        if (GROUPED) {
            const int xcd = (tile & 255) >> 5;
            int sq = (tile & 31) + 32 * (tile >> 8);
            // the e4m3 kernels walk statically: a workgroup's tiles of rounds 2u and 2u + 1 are made NEIGHBOURS (positions 2w, 2w + 1 of a
            // block of 64) -- what the paired ticket draws do for the bf16 kernels: two neighbouring column tiles of a row one after the
            // other by one workgroup instead of side by side by two.  ViT-L shape: c_fc + QuickGELU + emit 2130-2162 -> 2038-2055 us,
            // QuickGELU' + emit 2312 -> 2270 us, bit-identical.  An odd last round keeps its plain positions.  Bit 27: off (A/B).
            if (ES == 1 && !(p.dbg & (1 << 27))) {
                const int w = sq & 31, t = sq >> 5, rounds = ntiles >> 8;
                if ((t | 1) < rounds) sq = 64 * (t >> 1) + 2 * w + (t & 1);
            }
            const int pl = sq / cg;
            tm = (xcd & 3) * ppx + pl; tn = (xcd >> 2) * cg + sq % cg;
            valid = valid && pl < ppx && tm < ntm;
        }
        if (valid) {
            d.m0 = tm * BM; d.n0 = tn * BN;
            const int64_t a_bytes = ((int64_t)(p.M - d.m0) * p.lda - (p.lda - p.K)) * ES;
            const int64_t b_bytes = ((int64_t)(p.N - d.n0) * p.ldb - (p.ldb - p.K)) * ES;
            d.a = (const char*)p.A + (int64_t)d.m0 * p.lda * ES; d.abytes = (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes);
            d.b = (const char*)p.B + (int64_t)d.n0 * p.ldb * ES; d.bbytes = (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes);
        } else {                                   // past the last tile: zero records, every lane out of range, no traffic
            d.m0 = p.M; d.n0 = 0;
            d.a = (const char*)p.A; d.b = (const char*)p.B; d.abytes = 0; d.bbytes = 0;
        }
        return d;
    };
    auto fill_a = [&](const __amdgpu_buffer_rsrc_t& rs, int stage, uint32_t koff) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            lds_dma16(rs, ldsA + stage * PP_A_STAGE + i * 1024, voffA[i & 1], koff + waveA + (i >> 1) * pairA);
    };
    auto fill_b = [&](const __amdgpu_buffer_rsrc_t& rs, int slot, uint32_t koff) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            lds_dma16(rs, ldsB + slot * PP_B_SLOT + i * 1024, voffB[i & 1], koff + waveB + (i >> 1) * pairB);
    };
    // DEEP schedule (VAR 10; intervals by row halves, B fragments kept in registers): a wave owns 16 rows in EACH 64-row half of
    // its group's A stage (two 1-KiB pieces per half), so that a half can be refilled as soon as its interval is over
    constexpr bool DEEP = VAR == 10 || VAR == 12;
    char* const ldsA2 = smem + grp * (2 * PP_A_STAGE) + wl * 2048;
    const uint32_t waveA2 = (uint32_t)((grp * 128 + wl * 16) * p.lda * ES), halfA = (uint32_t)(64 * p.lda * ES);
    auto fill_a_half = [&](const __amdgpu_buffer_rsrc_t& rs, int stage, int half, uint32_t koff) {
        lds_dma16(rs, ldsA2 + stage * PP_A_STAGE + half * 8192, voffA[0], koff + waveA2 + half * halfA);
        lds_dma16(rs, ldsA2 + stage * PP_A_STAGE + half * 8192 + 1024, voffA[1], koff + waveA2 + half * halfA);
    };

    f32x4 acc[8][4];
    // Fragment pipeline of one 32-deep k-step (= one barrier interval): B fragments up front, A fragments two MFMA groups
    // ahead in a 3-deep ring.  (Reading k-step 1's first fragments ahead of the mid-K-tile barrier measured 5 % slower.)
    bf16x8 fb[4], fa[3];
    auto frag_head = [&](int ks, int stage, int slot) {
        const char* sa = smem + stage * PP_A_STAGE + offA[ks];
        const char* sb = smem + slot * PP_B_SLOT + offB[ks];
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(sb + j * 2048);
        fa[0] = *(const bf16x8*)(sa);
        fa[1] = *(const bf16x8*)(sa + 2048);
    };
    // 32 MFMAs of k-step ks
    auto half_body = [&](int ks, int stage) {
        const char* sa = smem + stage * PP_A_STAGE + offA[ks];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i + 2 < 8) fa[(i + 2) % 3] = *(const bf16x8*)(sa + (i + 2) * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i % 3], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // e4m3 operands: the K-tile is one 128-deep MFMA step; the two barrier intervals take row tiles 0-3 and 4-7 of the wave
    // (16 MFMAs of 32 cycles each, the length of the 32 bf16 MFMAs they replace).  The B fragments (both 16-byte halves of four
    // column tiles) are read in the first interval and kept; A fragments run one row tile (four MFMAs, 128 cycles) ahead.
    i32x8 gb[4], ga[2];
    uint32_t sav[2] = {0x7F7F7F7Fu, 0x7F7F7F7Fu}, sbv = 0x7F7F7F7Fu, sav_n[2] = {0x7F7F7F7Fu, 0x7F7F7F7Fu}, sbv_n = 0x7F7F7F7Fu;
    // B (weights): one exponent per row, column tile j -> byte j, once per tile.  A (activations): one exponent per row and 32 k (MX
    // layout, common.h): the bytes of this lane's eight row tiles for ONE K-tile are one 8-byte word, row tile i -> byte i, fetched a
    // K-tile ahead -- issued in front of the K-tile's DMA pieces, i.e. older than all of them, so that the counted waits at the end of
    // the K-tile cover it (the interval-0 waits allow one more request in flight for it).
    auto load_b_scales = [&](const TileDesc& d, uint32_t& b1) {
        b1 = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = d.n0 + wl * 64 + j * 16 + frow;
            b1 |= (uint32_t)(n < p.N ? p.sb[n] : (uint8_t)127) << (j * 8);
        }
    };
    const int kts = p.K >> 7;
    const int64_t groups = ((int64_t)p.M + 127) >> 7;
    auto load_a_scales = [&](const TileDesc& d, int kt, uint32_t (&a2)[2]) {
        const int64_t G = (int64_t)(d.m0 >> 7) + grp;
        u32x2 v = u32x2{0x7F7F7F7Fu, 0x7F7F7F7Fu};
        if (d.abytes != 0 && G < groups) v = *(const u32x2*)(p.sa + ((G * kts + kt) * 16 + frow) * 32 + fq * 8);
        a2[0] = v[0]; a2[1] = v[1];
    };
    auto frag_head8 = [&](auto hc, int stage, int slot) {
        constexpr int h = decltype(hc)::value;
        const char* sa0 = smem + stage * PP_A_STAGE + offA[0] + h * 4 * 2048;
        const char* sa1 = smem + stage * PP_A_STAGE + offA[1] + h * 4 * 2048;
        ga[0].lo = *(const i32x4*)(sa0);          // first: the first MFMA waits for these and gb[0] only (LDS returns in order)
        ga[0].hi = *(const i32x4*)(sa1);
        if (h == 0) {
            const char* sb0 = smem + slot * PP_B_SLOT + offB[0];
            const char* sb1 = smem + slot * PP_B_SLOT + offB[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gb[j].lo = *(const i32x4*)(sb0 + j * 2048);
                gb[j].hi = *(const i32x4*)(sb1 + j * 2048);
            }
        }
    };
    auto half_body8 = [&](auto hc, int stage) {
        constexpr int h = decltype(hc)::value;
        const char* sa0 = smem + stage * PP_A_STAGE + offA[0] + h * 4 * 2048;
        const char* sa1 = smem + stage * PP_A_STAGE + offA[1] + h * 4 * 2048;
        const uint32_t sah = sav[h];
        __builtin_amdgcn_sched_barrier(0);
#define VIPANT_MX_ROW(II)                                                                  \
        if (II + 1 < 4) {                                                                  \
            ga[(II + 1) & 1].lo = *(const i32x4*)(sa0 + (II + 1) * 2048);                  \
            ga[(II + 1) & 1].hi = *(const i32x4*)(sa1 + (II + 1) * 2048);                  \
        }                                                                                  \
        if (ES == 1) {                                                                     \
            VIPANT_MX(acc[h * 4 + II][0], gb[0], ga[II & 1], 0, sbv, II, sah);             \
            VIPANT_MX(acc[h * 4 + II][1], gb[1], ga[II & 1], 1, sbv, II, sah);             \
            VIPANT_MX(acc[h * 4 + II][2], gb[2], ga[II & 1], 2, sbv, II, sah);             \
            VIPANT_MX(acc[h * 4 + II][3], gb[3], ga[II & 1], 3, sbv, II, sah);             \
        } else {        /* bf16 under the DEEP schedule: k-step 0 of the four column tiles, then k-step 1 */ \
            VIPANT_BF(acc[h * 4 + II][0], gb[0].lo, ga[II & 1].lo); VIPANT_BF(acc[h * 4 + II][1], gb[1].lo, ga[II & 1].lo); \
            VIPANT_BF(acc[h * 4 + II][2], gb[2].lo, ga[II & 1].lo); VIPANT_BF(acc[h * 4 + II][3], gb[3].lo, ga[II & 1].lo); \
            VIPANT_BF(acc[h * 4 + II][0], gb[0].hi, ga[II & 1].hi); VIPANT_BF(acc[h * 4 + II][1], gb[1].hi, ga[II & 1].hi); \
            VIPANT_BF(acc[h * 4 + II][2], gb[2].hi, ga[II & 1].hi); VIPANT_BF(acc[h * 4 + II][3], gb[3].hi, ga[II & 1].hi); \
        }                                                                                  \
        __builtin_amdgcn_sched_barrier(0);
        VIPANT_MX_ROW(0) VIPANT_MX_ROW(1) VIPANT_MX_ROW(2) VIPANT_MX_ROW(3)
#undef VIPANT_MX_ROW
    };

    int tile = lane_pos, tile_nxt = lane_pos + G;
    TileDesc cur = describe(tile), nxt = describe(tile_nxt);
    int gk = 0, slot = 0;                       // K-tile counter of the stream: A stage = gk & 1, B slot = gk % 3
    if (ES == 1) {
        load_a_scales(cur, 0, sav);
        load_b_scales(cur, sbv);
        asm volatile("" : "+v"(sav[0]), "+v"(sav[1]), "+v"(sbv));             // awaited before any DMA is in flight
    }
    if (DEEP) {     // K-tile 0 (A, B), K-tile 1 (B; A rows 0-63): what the steady state would have issued before K-tile 0
        const bool wrap = nk < 2;
        const __amdgpu_buffer_rsrc_t ra1 = wrap ? make_rsrc(nxt.a, nxt.abytes) : make_rsrc(cur.a, cur.abytes);
        const __amdgpu_buffer_rsrc_t rb1 = wrap ? make_rsrc(nxt.b, nxt.bbytes) : make_rsrc(cur.b, cur.bbytes);
        fill_a_half(make_rsrc(cur.a, cur.abytes), 0, 0, 0);
        fill_a_half(make_rsrc(cur.a, cur.abytes), 0, 1, 0);
        fill_b(make_rsrc(cur.b, cur.bbytes), 0, 0);
        fill_b(rb1, 1, wrap ? 0u : (uint32_t)(BK * 2));
        fill_a_half(ra1, 1, 0, wrap ? 0u : (uint32_t)(BK * 2));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // prologue: K-tile 0 of the first tile (and, for group 1, its B rows of K-tile 1)
    fill_a(make_rsrc(cur.a, cur.abytes), 0, 0);
    fill_b(make_rsrc(cur.b, cur.bbytes), 0, 0);
    if (grp == 1) {
        const bool wrap = nk < 2;
        fill_b(wrap ? make_rsrc(nxt.b, nxt.bbytes) : make_rsrc(cur.b, cur.bbytes), 1, wrap ? 0u : (uint32_t)(BK * 2));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();         // the lag: group 1 runs one barrier interval behind group 0

    // (tk_post, the data register of the mailbox store, lives across the whole walk -- the kernel's last statement reads it.  hipcc
    // protects a register that a store in flight still names with a vmcnt wait in front of its next writer: a wait behind LDS-DMA
    // pieces or code loads wherever the allocator happened to reuse it.)
    uint32_t tk_post = 0u;
    uint32_t tk_raw = 0u;                       // the mailbox word in flight (asm load; read only behind the next K-tile 0's counted waits)
    bool tk_defer = false;
    int tile_nn = tile_nxt + G;                 // the tile after `tile_nxt` (ticket walk: static for a workgroup's first four tiles)
    while (tile < ntiles) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 bv[4];
        constexpr bool PRE_CODES = DEEP && EPI == VIPANT_EPI_DQUICKGELU_D8;
        u32x2 cn_pre[4];
        for (int k = 0; k < nk; ++k) {
            const int stage = gk & 1;
            const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
            if (k == nk - 1) {                   // the tile's bias: fetched and awaited before this K-tile's DMA is queued
                load_bias<EPI>(p, cur.n0, wl, fq, bv);
                if (ES == 1) load_b_scales(nxt, sbv_n);                       // the next tile's weight scales ride the same round trip
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bv[j]));
                if (ES == 1) asm volatile("" : "+v"(sbv_n));
                if (dyn) {
                    // the ticket drawn one epilogue ago has returned with the loads above: publish it for the next tile's round trip
                    tk_post = (uint32_t)(tk_dry ? NO_TILE : tk_tile(tk_pend + (uint32_t)tk_half));
                    if (tk_n == 2) tk_half ^= 1;     // (1: the second position of this draw is next, the coming epilogue draws nothing)
                    if (tk_taper && tk_half == 0 && tk_n == 2 && 128 + (int)tk_pend + 96 >= qlen_own) tk_n = 1;
                    tk_par ^= 1;                 // (the store itself goes out at the start of the epilogue, in front of the tile's own stores)
                    // the mailbox word requested behind the previous epilogue (asm load, see there): nk - 1 >= 3 K-tiles of counted
                    // waits have retired it.  It names the tile after `tile_nxt`; a scalar from here.
                    asm volatile("" : "+v"(tk_raw));
                    const int drawn = __builtin_amdgcn_readfirstlane((int)tk_raw);
                    tile_nn = tk_defer ? drawn : tile_nn;
                    asm volatile("" : "+v"(tk_par));
                }
                // (issued behind the waits above -- the code bytes come from HBM -- and not awaited here)
                if (PRE_CODES) load_codes(p, cur.m0, cur.n0, grp, tid & 255, 0, cn_pre);
            }
            if (ES == 1) {      // the A scales of the stream's next K-tile
                const bool w1 = k + 1 >= nk;
                load_a_scales(w1 ? nxt : cur, w1 ? 0 : k + 1, sav_n);
            }
            if (DEEP) {
                // interval 0 (row tiles 0-3): B rows of K-tile k+2 (its slot was last read one interval ago by the lagging group),
                // A rows 64-127 of K-tile k+1 (their half-stage was consumed in the previous interval); interval 1 (row tiles 4-7):
                // A rows 0-63 of K-tile k+2.  Every piece has three intervals (B of group 0: four) to land; at most the eight most
                // recent pieces may be outstanding at either barrier.
                const bool w1 = k + 1 >= nk, w2 = k + 2 >= nk;
                const uint32_t k1 = (uint32_t)((w1 ? k + 1 - nk : k + 1) * BK * 2), k2 = (uint32_t)((w2 ? k + 2 - nk : k + 2) * BK * 2);
                fill_b(make_rsrc(w2 ? nxt.b : cur.b, w2 ? nxt.bbytes : cur.bbytes), slot2, k2);
                fill_a_half(make_rsrc(w1 ? nxt.a : cur.a, w1 ? nxt.abytes : cur.abytes), stage ^ 1, 1, k1);
                __builtin_amdgcn_sched_barrier(0);
                frag_head8(Int<0>{}, stage, slot);
                half_body8(Int<0>{}, stage);
                if (ES == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");       // (+ the A-scale word of the next K-tile)
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                sync();
                fill_a_half(make_rsrc(w2 ? nxt.a : cur.a, w2 ? nxt.abytes : cur.abytes), stage, 0, k2);
                __builtin_amdgcn_sched_barrier(0);
                frag_head8(Int<1>{}, stage, slot);
                half_body8(Int<1>{}, stage);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                sync();
            } else
            {   // first interval: DMA of the stream's next K-tile(s), k-step 0
                const bool w1 = k + 1 >= nk, w2 = k + 2 >= nk;
                const bool wb = grp == 0 ? w1 : w2;
                const __amdgpu_buffer_rsrc_t ra = make_rsrc(w1 ? nxt.a : cur.a, w1 ? nxt.abytes : cur.abytes);
                const __amdgpu_buffer_rsrc_t rb = make_rsrc(wb ? nxt.b : cur.b, wb ? nxt.bbytes : cur.bbytes);
                const uint32_t ka = w1 ? 0u : (uint32_t)((k + 1) * BK * 2);
                const uint32_t kb = grp == 0 ? ka : (uint32_t)((w2 ? k + 2 - nk : k + 2) * BK * 2);
                char* const da = ldsA + (stage ^ 1) * PP_A_STAGE;
                char* const db = ldsB + (grp == 0 ? slot1 : slot2) * PP_B_SLOT;
                // the K-tile's eight pieces as one burst at its top: 4 of A, then 4 of B.  (Measured and not kept, see
                // profiles/r2_gemm_experiments.md: fragment reads before the burst, the pieces spread behind the MFMA groups, raised
                // MFMA priority, `nt` hints on either operand, the burst half a K-tile later.)
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_dma16(ra, da + i * 1024, voffA[i & 1], ka + waveA + (i >> 1) * pairA);
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_dma16(rb, db + i * 1024, voffB[i & 1], kb + waveB + (i >> 1) * pairB);
                __builtin_amdgcn_sched_barrier(0);
                if (ES == 1) frag_head8(Int<0>{}, stage, slot); else
                frag_head(0, stage, slot);
                if (ES == 1) half_body8(Int<0>{}, stage); else
                half_body(0, stage);
                if (grp == 1) {
                    if (ES == 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // (+ the A-scale word of the next K-tile)
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
                sync();
                // second interval: k-step 1
                if (ES == 1) frag_head8(Int<1>{}, stage, slot); else
                frag_head(1, stage, slot);
                if (ES == 1) half_body8(Int<1>{}, stage); else
                half_body(1, stage);
                if (grp == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sync();
            }
            ++gk;
            slot = slot1;
            if (ES == 1) { sav[0] = sav_n[0]; sav[1] = sav_n[1]; }
        }
        // epilogue: 4 rounds of 32 rows through the A stage this group's stream does not use (the one just read)
        char* stg = smem + grp * (2 * PP_A_STAGE) + ((gk - 1) & 1) * PP_A_STAGE;
        if (DEEP) stg = smem + PP_B_BASE + (slot == 0 ? 2 : slot - 1) * PP_B_SLOT + grp * 16384;   // the last K-tile's B slot: read and done
        // the mailbox store goes here and not next to the bias: there it was the first operation behind a full wait, and the next
        // counted wait -- which lets the K-tile's eight pieces stay in flight -- had to see its acknowledgement first (+0.5 us per tile
        // on the launches with a bias); here the tile's own stores queue up behind it and nobody waits for it in particular
        // (What the draw still costs, round 5, bits switched off one by one in a test build: nothing measurable on the launches whose
        // epilogue only stores; 10-25 us per launch on QuickGELU', whose epilogue waits for its code loads every round -- vmcnt retires
        // in order, so thread 0's wave sees those loads only when the atomic, a ~2 us round trip, is back too, wherever in the
        // epilogue it is issued, and the other waves meet it at the round's barrier.)
        if (dyn && tid == 0) {
            tickets::post_g(mbox + 4 * tk_par, tk_post);
            if (!tk_dry && !tk_half) tk_pend = tickets::take_g(tkq, (uint32_t)tk_n);      // not awaited here
        }
        // (safe for the e4m3 kernels only: their B fragments are read in a K-tile's FIRST interval and kept, so when the leading group
        // starts its epilogue the lagging group, one interval behind, is past its last read of that slot)
        static_assert(EMIT == 0 || (!DEEP && ES == 1), "the e4m3 form's scale words live in the B slot of the last K-tile");
        pp_epilogue<EPI, EMIT>(p, acc, bv, stg, stg + 16 * 512, cur.m0, cur.n0, grp, wl, frow, fq, tid, PRE_CODES ? &cn_pre : nullptr,
                               (uint32_t*)(smem + PP_B_BASE + (slot == 0 ? 2 : slot - 1) * PP_B_SLOT) + grp * 32);
        // Every wave requests the word posted ONE epilogue ago (the other parity): the index of the tile after the next.  The request
        // is inline asm: a load hipcc can see would be awaited where it is first used with a wait it cannot count (the epilogue's stores
        // sit behind bounds checks: it would be vmcnt(0), every store's acknowledgement), and anywhere inside the K-loop a load has one
        // barrier interval to return (that was the first form: read at K-tile nk - 2, +0.5 us per tile on several launches).  Issued
        // here, behind the tile's stores, it costs no wait of its own: the counted waits of the next K-tile 0 retire those stores anyway
        // and the load with them (it is older than that K-tile's eight pieces); only then is the register read.  (An accumulation
        // register as destination is not affordable: one AGPR costs a granule of eight registers and the K-loop spills.)
        if (dyn) asm volatile("global_load_dword %0, %1, off" : "=&v"(tk_raw) : "v"(mbox + 4 * (tk_par ^ 1)) : "memory");
        tile = tile_nxt;
        cur = nxt;
        if (dyn) {
            // the word requested above is this tile's successor's SUCCESSOR's successor: it becomes `tile_nn` at the end of the new
            // tile's K-tile 0 (one scalar; describing a tile costs registers the K-loop does not have) and `nxt` one epilogue later
            tile_nxt = tile_nn;
            tk_defer = !__builtin_amdgcn_readfirstlane(tk_first);
            if (!tk_defer) tile_nn = tile_nxt + G;          // after the first epilogue nothing has been posted yet: the fourth static tile
        } else {
            tile_nxt = tile_nxt + G;
        }
        nxt = describe(tile_nxt);
        if (DYN) { tk_first = 0; asm volatile("" : "+v"(tk_first)); }
        if (ES == 1) sbv = sbv_n;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();         // pairs with group 1's last barrier
    if (dyn && tk_post == 0xA5A5A5A5u) tickets::post_g(mbox, tk_post);       // never true (tickets are < 2^30): keeps the store's data register
}

template <int EPI, int VAR, int ES = 2, int EMIT = 0>
int32_t launch_pp_variant(const GemmNT& p_in, hipStream_t stream) {
    // the ticket walk exists for the bf16 kernels (the e4m3 ones sit at the 256-register limit: BASELINE configs[4] keeps the static walk)
    constexpr bool CAN_DYN = ES == 2;
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, VAR, ES, EMIT, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
        if (CAN_DYN)
            VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_pp_kernel<EPI, VAR, ES, EMIT, CAN_DYN>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));
        done_on_device(once);
    }
    GemmNT p = p_in;
    const int64_t ntm = ceil_div(p.M, BM), ntn = ceil_div(p.N, BN);
    const int64_t tiles = ntm * ntn;
    int64_t grid = tiles < 256 ? (tiles + 7) / 8 * 8 : 256;
    // ticket walk (common.h) when every XCD's queue holds more than the three rounds a workgroup takes statically: the shortest queue
    // of the plain walk is the last one, of the column-grouped walk (VAR 8) the one of the last row quarter
    const int64_t ppx = (ntm + 3) / 4;
    const int64_t shortest = (VAR == 8 || VAR == 12) ? (ntm - 3 * ppx > 0 ? (ntm - 3 * ppx < ppx ? ntm - 3 * ppx : ppx) : 0) * (ntn / 2)
                                      : (tiles >> 8) * 32 + ((tiles & 255) > 224 ? (tiles & 255) - 224 : 0);
    if (CAN_DYN && grid == 256 && shortest > 128 && p.K >= 4 * BK && !(p.dbg & 4194304)) {      // bit 22 of VIPANT_GEMM_VARIANT: static walk (A/B)
        p.tk = vipant_ticket_block(stream, &p.tk_other);
        if (p.tk == nullptr) return VIPANT_EHIP;
        hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, VAR, ES, EMIT, CAN_DYN>), dim3((unsigned)grid), dim3(512), PP_LDS_BYTES, stream, p);
        VIPANT_LAUNCH_CHECK();
        return VIPANT_OK;
    }
    hipLaunchKernelGGL((gemm_nt_pp_kernel<EPI, VAR, ES, EMIT, false>), dim3((unsigned)grid), dim3(512), PP_LDS_BYTES, stream, p);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}


template <int EPI>
int32_t launch_pp(const GemmNT& p, hipStream_t stream) {
    // schedules of the ping-pong kernel: 0 = k-step intervals, 8 = the same on the column-grouped tile walk, 10 = DEEP, 12 = DEEP on the grouped walk
    if (p.dbg & 131072) return launch_pp_variant<EPI, 10>(p, stream);      // bit 17: DEEP for every launch (A/B)
    const bool groupable = ceil_div(p.N, BN) % 2 == 0 && ceil_div(p.M, BM) * ceil_div(p.N, BN) >= 256;
    // the column-grouped walk is the default of the c_fc launch (853 vs 870-881 us, step -0.27 ms in-box; the QuickGELU' launch of
    // the same shape does not move: profiles/r2_gemm_experiments.md section 9); bit 11 forces it everywhere, bit 12 turns it off
    // round 5: the two K = 768 launches with 12 column tiles (c_fc + QuickGELU, QuickGELU') run the DEEP schedule ON the column-grouped
    // walk (variant 12): half of the weight matrix per XCD stays in its L2 (the plain DEEP walk re-streams all of it every round:
    // QuickGELU' fetched 1.93 GB for 0.75 GB of operands) and the look-ahead of DEEP is kept; 878.9 -> 870.2 us and 883.4 -> 872.2 us,
    // alternating in one process, bit-identical (tools/grouped_walk_ab.py).  Bit 23 of VIPANT_GEMM_VARIANT: the round-4 choice (8 / 10).
    if constexpr (EPI == VIPANT_EPI_QUICKGELU_D8 || EPI == VIPANT_EPI_DQUICKGELU_D8)
        if (groupable && !(p.dbg & (8388608 | 4096 | 2048 | 262144))) return launch_pp_variant<EPI, 12>(p, stream);
    // ... and plain launches with >= 8 column tiles (no shape of the ViT-B step: qkv has 9; the ViT-L qkv launch has 12)
    if constexpr (EPI == VIPANT_EPI_BF16)
        if (groupable && ceil_div(p.N, BN) >= 8 && !(p.dbg & (8388608 | 4096 | 2048 | 262144))) return launch_pp_variant<EPI, 12>(p, stream);
    if (groupable && !(p.dbg & 4096) && (EPI == VIPANT_EPI_QUICKGELU_D8 || (p.dbg & 2048))) return launch_pp_variant<EPI, 8>(p, stream);
    // the DEEP schedule (three barrier intervals of look-ahead for every operand piece, intervals by row halves): -2 ... -5 % on the
    // launches with a long K or a wide N (qkv 608-624 -> 589-603 us, QuickGELU' 933-947 -> 916, dh2 627-634 -> 596-604), +3 % on the
    // 768 x 768 ones, which keep the k-step schedule; bit 18 of VIPANT_GEMM_VARIANT: off
    if (!(p.dbg & 262144) && (EPI == VIPANT_EPI_DQUICKGELU_D8 || (EPI == VIPANT_EPI_BF16 && (p.N >= 1024 || p.K >= 1024))))
        return launch_pp_variant<EPI, 10>(p, stream);
    return launch_pp_variant<EPI, 0>(p, stream);
}

template <int EPI>
int32_t launch_persistent(const GemmNT& p, hipStream_t stream) {
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_persistent_kernel<EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        done_on_device(once);
    }
    const int64_t tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    int64_t grid = tiles < 256 ? (tiles + 7) / 8 * 8 : 256;
    hipLaunchKernelGGL(gemm_nt_persistent_kernel<EPI>, dim3((unsigned)grid), dim3(512), 2 * STAGE_BYTES, stream, p);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int EPI>
int32_t launch(const GemmNT& p, hipStream_t stream) {
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        done_on_device(once);
    }
    const int64_t tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES, stream, p);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// Few rows (VIPANT_EPI_FEW_ROWS): the last block on its read-out rows (`batch` rows per launch) and the read-out.
// A 256 x 256 tile leaves such a launch on 6-24 CUs for 12-48 serial K-steps (21-91 us measured at M = 512).  Here: 64 x 64 tiles
// (96-384 workgroups), the four waves of a workgroup split K between them -- fragments straight from global memory (both
// operands are K-contiguous, 16 bytes per lane and MFMA operand, no LDS stage) with a whole batch of K-steps in flight -- and add
// their partial tiles through LDS; wave w then owns rows 16 w .. 16 w + 15 of the tile for the epilogue (4 consecutive columns of
// one row per lane, as everywhere in this file).
constexpr int SK_KB = 6;          // K-steps (of 32) a wave has in flight
template <int EPI, int NW, bool PAIR = false>     // NW waves share K.  Four: eight were tried for K = 3072 and are slower (34 against 30 us at M = 512)
__global__ __launch_bounds__(NW * 64) void gemm_nt_skinny_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* part = (f32x4*)smem;                           // [NW waves][16 tiles][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int ntn = (p.N + 63) / 64;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * 64, n0 = tn * 64;
    const int64_t z = blockIdx.y;                            // (vipant_gemm_nt_heads: one product per head)
    const bf16_t *ap[4], *bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * i + r, n = n0 + 16 * i + r;
        ap[i] = p.A + z * p.stride_a + (int64_t)(m < p.M ? m : p.M - 1) * p.lda + 8 * g;
        bp[i] = p.B + z * p.stride_b + (int64_t)(n < p.N ? n : p.N - 1) * p.ldb + 8 * g;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk1 = p.K / 32;
    const int nks = (PAIR && p.a_lo != 0) ? 2 * nk1 : nk1;        // a pair in A: K-steps nk1 .. 2 nk1 - 1 take the lo plane
    for (int s0 = wave * SK_KB; s0 < nks; s0 += NW * SK_KB) {
        bf16x8 af[SK_KB][4], bf[SK_KB][4];
#pragma unroll
        for (int t = 0; t < SK_KB; ++t) {
            const int st = s0 + t < nks ? s0 + t : nks - 1;
            const int k = (PAIR && st >= nk1 ? st - nk1 : st) * 32;
            const int64_t ka = PAIR && st >= nk1 ? p.a_lo + k : (int64_t)k;
#pragma unroll
            for (int i = 0; i < 4; ++i) { af[t][i] = *(const bf16x8*)(ap[i] + ka); bf[t][i] = *(const bf16x8*)(bp[i] + k); }
        }
        __builtin_amdgcn_sched_barrier(0);     // all of the batch's loads issued before the first MFMA waits for one
#pragma unroll
        for (int t = 0; t < SK_KB; ++t) {
            if (s0 + t >= nks) break;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[t][j], af[t][i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) part[(wave * 16 + i * 4 + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    // wave w takes rows 16 (w & 3) .. of the tile; with eight waves, waves 4-7 the right half of the columns
    const int mi = wave & 3;
    const int m = m0 + 16 * mi + r;
#pragma unroll
    for (int jj = 0; jj < 16 / NW; ++jj) {
        const int j = NW == 8 ? 2 * (wave >> 2) + jj : jj;
        const int n4 = n0 + 16 * j + 4 * g;
        f32x4 v = part[(mi * 4 + j) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v += part[(ww * 16 + mi * 4 + j) * 64 + lane];
        if (m >= p.M || n4 >= p.N) continue;
        const int64_t o = z * p.stride_c + (int64_t)m * p.ldc + n4;
        if (EPI != VIPANT_EPI_DQUICKGELU_D8 && p.bias != nullptr) v += *(const f32x4*)(p.bias + z * p.stride_bias + n4);
        if (EPI == VIPANT_EPI_BF16) {
            const bf16x4 hi = f32x4_to_bf16x4(v);
            *(bf16x4*)((bf16_t*)p.C + o) = hi;
            if (PAIR && p.c_lo != 0) {
                f32x4 rest;
#pragma unroll
                for (int e = 0; e < 4; ++e) rest[e] = v[e] - (float)hi[e];
                *(bf16x4*)((bf16_t*)p.C + p.c_lo + o) = f32x4_to_bf16x4(rest);
            }
        } else if (EPI == VIPANT_EPI_F32) {
            *(f32x4*)((float*)p.C + o) = v;
        } else if (EPI == VIPANT_EPI_RESIDUAL_F32) {
            *(f32x4*)((float*)p.C + o) = v + *(const f32x4*)((const float*)p.aux + o);
        } else if (EPI == VIPANT_EPI_QUICKGELU_D8) {      // as the big kernel: the gate sees the bf16-rounded pre-activation
            const bf16x4 ub = f32x4_to_bf16x4(v);
            f32x4 ge;
            uint32_t code = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = (float)ub[e];
                const float sg = quickgelu_gate(u);
                ge[e] = u * sg;
                code = gelu_code_pack(sg * (1.0f + 1.702f * u * (1.0f - sg)), e, code);
            }
            *(bf16x4*)((bf16_t*)p.C + o) = f32x4_to_bf16x4(ge);
            *(uint32_t*)((uint8_t*)p.aux + o) = code;
        } else {  // VIPANT_EPI_DQUICKGELU_D8
            const uint32_t code = *(const uint32_t*)((const uint8_t*)p.aux + o);
            f32x4 d;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = v[e] * gelu_decode((code >> (8 * e)) & 255u);
            *(bf16x4*)((bf16_t*)p.C + o) = f32x4_to_bf16x4(d);
        }
    }
}

template <int EPI, int NW, bool PAIR = false>
int32_t launch_skinny_nw(const GemmNT& p, hipStream_t stream, int nb = 1) {
    constexpr int lds = NW * 16 * 64 * 16;
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_skinny_kernel<EPI, NW, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        done_on_device(once);
    }
    const unsigned grid = (unsigned)(((p.M + 63) / 64) * ((p.N + 63) / 64));
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, NW, PAIR>), dim3(grid, (unsigned)nb), dim3(NW * 64), lds, stream, p);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}
template <int EPI>
int32_t launch_skinny(const GemmNT& p, hipStream_t stream) {
    return launch_skinny_nw<EPI, 4>(p, stream);
}

extern "C" int32_t vipant_gemm_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C,
                                  int64_t ldc, const float* bias, void* aux, float alpha, int64_t M, int64_t N,
                                  int64_t K, int32_t epilogue, void* stream) {
    VIPANT_REQUIRE(M > 0 && N > 0 && K > 0, VIPANT_EBADSHAPE, "gemm_nt: empty problem M=%ld N=%ld K=%ld",
                   (long)M, (long)N, (long)K);
    VIPANT_REQUIRE(K % 64 == 0 && N % 4 == 0, VIPANT_EBADSHAPE, "gemm_nt: need K%%64==0 and N%%4==0 (K=%ld N=%ld)",
                   (long)K, (long)N);
    VIPANT_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, VIPANT_EALIGN,
                   "gemm_nt: bad leading dims lda=%ld ldb=%ld ldc=%ld", (long)lda, (long)ldb, (long)ldc);
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), VIPANT_EALIGN,
                   "gemm_nt: operands must be 16-byte aligned");
    VIPANT_REQUIRE(256 * lda * 2 < (1ll << 31) && 256 * ldb * 2 < (1ll << 31), VIPANT_EBADSHAPE,
                   "gemm_nt: leading dimension too large");
    const char* var = getenv("VIPANT_GEMM_VARIANT");       // read per call: tests and A/B scripts switch it inside one process
    const int dbg = var ? atoi(var) : 0;
    GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, bias, aux, lda, ldb, ldc, (int)M, (int)N, (int)K, alpha, dbg, nullptr, nullptr, 0, nullptr};
    hipStream_t s = (hipStream_t)stream;
    const bool staged = (N % 8 == 0) && (ldc % 8 == 0) && !(dbg & 2);
    const bool pp = staged && K >= 128 && !(dbg & 16);
    const bool few_rows = (epilogue & VIPANT_EPI_FEW_ROWS) != 0;
    epilogue &= ~VIPANT_EPI_FEW_ROWS;
    if (few_rows && !(dbg & 2097152)) {        // bit 21 of VIPANT_GEMM_VARIANT sends them through the 256 x 256 kernels again (A/B)
        switch (epilogue) {
            case VIPANT_EPI_BF16: return launch_skinny<VIPANT_EPI_BF16>(p, s);
            case VIPANT_EPI_F32: return launch_skinny<VIPANT_EPI_F32>(p, s);
            case VIPANT_EPI_RESIDUAL_F32:
                VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: residual epilogue needs aux");
                return launch_skinny<VIPANT_EPI_RESIDUAL_F32>(p, s);
            case VIPANT_EPI_QUICKGELU_D8:
            case VIPANT_EPI_DQUICKGELU_D8:
                VIPANT_REQUIRE(aux != nullptr && (uintptr_t)aux % 16 == 0, VIPANT_EBADSHAPE,
                               "gemm_nt: the 8-bit QuickGELU' epilogues need aux (the code matrix), 16-byte aligned");
                return epilogue == VIPANT_EPI_QUICKGELU_D8 ? launch_skinny<VIPANT_EPI_QUICKGELU_D8>(p, s)
                                                            : launch_skinny<VIPANT_EPI_DQUICKGELU_D8>(p, s);
            default:
                vipant_set_error("gemm_nt: VIPANT_EPI_FEW_ROWS with epilogue %d (BF16, F32, RESIDUAL_F32 and the two _D8 epilogues only)", epilogue);
                return VIPANT_EBADSHAPE;
        }
    }
    switch (epilogue) {
        case VIPANT_EPI_BF16:
            if (pp) return launch_pp<VIPANT_EPI_BF16>(p, s);
            return staged ? launch_persistent<VIPANT_EPI_BF16>(p, s) : launch<VIPANT_EPI_BF16>(p, s);
        case VIPANT_EPI_F32: return launch<VIPANT_EPI_F32>(p, s);
        case VIPANT_EPI_RESIDUAL_F32:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: residual epilogue needs aux");
            // one tile per workgroup: the only caller on the step's path is the last block on its read-out rows (`batch` rows, a handful
            // of tiles), and the persistent form of this epilogue spilled 42 registers (the fp32 residual tile rides beside the
            // accumulators); the instantiation is gone, so nothing on the step's path can pick it up again
            return launch<VIPANT_EPI_RESIDUAL_F32>(p, s);
        case VIPANT_EPI_QUICKGELU:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: quickgelu epilogue needs aux (U out)");
            if (pp) return launch_pp<VIPANT_EPI_QUICKGELU>(p, s);
            return staged ? launch_persistent<VIPANT_EPI_QUICKGELU>(p, s) : launch<VIPANT_EPI_QUICKGELU>(p, s);
        case VIPANT_EPI_DQUICKGELU:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: dquickgelu epilogue needs aux (U in)");
            if (pp) return launch_pp<VIPANT_EPI_DQUICKGELU>(p, s);
            return staged ? launch_persistent<VIPANT_EPI_DQUICKGELU>(p, s) : launch<VIPANT_EPI_DQUICKGELU>(p, s);
        case VIPANT_EPI_QUICKGELU_D8:
        case VIPANT_EPI_DQUICKGELU_D8:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: the 8-bit QuickGELU' epilogues need aux (the code matrix)");
            VIPANT_REQUIRE(staged && K >= 128 && (uintptr_t)aux % 16 == 0, VIPANT_EBADSHAPE,
                           "gemm_nt: the 8-bit QuickGELU' epilogues need N %% 8 == 0, ldc %% 8 == 0, K >= 128 and a 16-byte aligned aux");
            return epilogue == VIPANT_EPI_QUICKGELU_D8 ? launch_pp<VIPANT_EPI_QUICKGELU_D8>(p, s)
                                                        : launch_pp<VIPANT_EPI_DQUICKGELU_D8>(p, s);
        case VIPANT_EPI_SCALE_F32: return launch<VIPANT_EPI_SCALE_F32>(p, s);
        default:
            vipant_set_error("gemm_nt: unknown epilogue %d", epilogue);
            return VIPANT_EBADSHAPE;
    }
}

// H independent small products in one launch of the few-rows kernel: the per-head contractions of the folded last block
// (csrc/readout_ctx.hip) -- a head's 64 columns of the `batch` read-out rows against the head's block of a weight matrix -- without
// materialising the block-sparse [batch * H, D] operand (vipant_head_expand) or computing its zeros.
extern "C" int32_t vipant_gemm_nt_heads(const uint16_t* A, int64_t lda, int64_t stride_a, int64_t a_lo, const uint16_t* B, int64_t ldb,
                                        int64_t stride_b, uint16_t* C, int64_t ldc, int64_t stride_c, int64_t c_lo, const float* bias,
                                        int64_t stride_bias, int64_t M, int64_t N, int64_t K, int64_t H, void* stream) {
    VIPANT_REQUIRE(M > 0 && N > 0 && K > 0 && H > 0 && H < 65536 && K % 64 == 0 && N % 4 == 0, VIPANT_EBADSHAPE,
                   "gemm_nt_heads: bad shape M=%ld N=%ld K=%ld H=%ld (K %% 64, N %% 4)", (long)M, (long)N, (long)K, (long)H);
    VIPANT_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && stride_a % 8 == 0 &&
                   stride_b % 8 == 0 && stride_c % 4 == 0 && stride_bias % 4 == 0, VIPANT_EALIGN,
                   "gemm_nt_heads: leading dimensions / head strides must keep 16-byte (operands), 8-byte (C) and 16-byte (bias) alignment");
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 8 == 0) && ((uintptr_t)bias % 16 == 0),
                   VIPANT_EALIGN, "gemm_nt_heads: operands must be 16-byte aligned");
    VIPANT_REQUIRE(a_lo % 8 == 0 && c_lo % 4 == 0 && a_lo >= 0 && c_lo >= 0, VIPANT_EALIGN, "gemm_nt_heads: the lo planes must keep the operands' alignment");
    GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, bias, nullptr, lda, ldb, ldc, (int)M, (int)N, (int)K, 1.0f, 0, nullptr, nullptr, 0, nullptr,
             stride_a, stride_b, stride_c, stride_bias};
    p.a_lo = a_lo; p.c_lo = c_lo;
    if (a_lo != 0 || c_lo != 0) return launch_skinny_nw<VIPANT_EPI_BF16, 4, true>(p, (hipStream_t)stream, (int)H);
    return launch_skinny_nw<VIPANT_EPI_BF16, 4>(p, (hipStream_t)stream, (int)H);
}

// The patch embedding written straight into the token matrix: tokens[item * (P + 1) + patch + 1, :] = A[item * P + patch, :] . B^T +
// pos[patch + 1, :], fp32 (the class-token rows are vipant_tokens_cls_rows').  Replaces an fp32 [b P, N] intermediate and the pass
// that re-read it (assemble_tokens): 0.22 ms of the step at cfg2.
extern "C" int32_t vipant_gemm_nt_tokens(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* tokens,
                                         const float* pos, int64_t b, int64_t P, int64_t N, int64_t K, void* stream) {
    VIPANT_REQUIRE(b > 0 && P > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0 && b * P < (1ll << 31), VIPANT_EBADSHAPE,
                   "gemm_nt_tokens: bad shape b=%ld P=%ld N=%ld K=%ld", (long)b, (long)P, (long)N, (long)K);
    VIPANT_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && 256 * lda * 2 < (1ll << 31) && 256 * ldb * 2 < (1ll << 31),
                   VIPANT_EALIGN, "gemm_nt_tokens: bad leading dims");
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)tokens % 16 == 0) && ((uintptr_t)pos % 16 == 0),
                   VIPANT_EALIGN, "gemm_nt_tokens: operands must be 16-byte aligned");
    GemmNT p{(const bf16_t*)A, (const bf16_t*)B, tokens, nullptr, nullptr, lda, ldb, N, (int)(b * P), (int)N, (int)K, 1.0f, 0, nullptr,
             nullptr, (int)P, pos};
    return launch<VIPANT_EPI_F32>(p, (hipStream_t)stream);
}

// e4m3 x e4m3 -> bf16: C = (A * 2^(sa - 127)) (B * 2^(sb - 127))^T [+ bias], the ping-pong kernel at ES = 1.
extern "C" int32_t vipant_gemm_nt_e4m3(const uint8_t* A, int64_t lda, const uint8_t* sa, const uint8_t* B, int64_t ldb,
                                       const uint8_t* sb, void* C, int64_t ldc, const float* bias, void* aux, uint8_t* cq, uint8_t* cq_scale,
                                       int64_t M, int64_t N, int64_t K, int32_t epilogue, void* stream) {
    VIPANT_REQUIRE(M > 0 && N > 0 && K > 0, VIPANT_EBADSHAPE, "gemm_nt_e4m3: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    VIPANT_REQUIRE(K % 128 == 0 && K >= 256 && N % 8 == 0, VIPANT_EBADSHAPE,
                   "gemm_nt_e4m3: need K %% 128 == 0, K >= 256 and N %% 8 == 0 (K=%ld N=%ld)", (long)K, (long)N);
    VIPANT_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 16 == 0 && ldb % 16 == 0 && ldc % 8 == 0, VIPANT_EALIGN,
                   "gemm_nt_e4m3: bad leading dims lda=%ld ldb=%ld ldc=%ld", (long)lda, (long)ldb, (long)ldc);
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), VIPANT_EALIGN,
                   "gemm_nt_e4m3: operands must be 16-byte aligned");
    VIPANT_REQUIRE(sa != nullptr && sb != nullptr && (uintptr_t)sa % 8 == 0, VIPANT_EBADSHAPE,
                   "gemm_nt_e4m3: the block scales of A (8-byte aligned) and the row scales of B are required");
    VIPANT_REQUIRE((cq == nullptr) == (cq_scale == nullptr), VIPANT_EBADSHAPE, "gemm_nt_e4m3: cq and cq_scale go together");
    VIPANT_REQUIRE(cq == nullptr || (N % 128 == 0 && (uintptr_t)cq % 16 == 0 && epilogue != VIPANT_EPI_BF16), VIPANT_EBADSHAPE,
                   "gemm_nt_e4m3: the e4m3 form of the result needs N %% 128 == 0, a 16-byte aligned cq and one of the QuickGELU epilogues");
    VIPANT_REQUIRE(C != nullptr || (cq != nullptr && (epilogue == VIPANT_EPI_QUICKGELU_D8 || epilogue == VIPANT_EPI_DQUICKGELU_D8)), VIPANT_EBADSHAPE,
                   "gemm_nt_e4m3: C may be NULL only when one of the QuickGELU epilogues leaves the e4m3 form in its place");
    VIPANT_REQUIRE(256 * lda < (1ll << 31) && 256 * ldb < (1ll << 31), VIPANT_EBADSHAPE, "gemm_nt_e4m3: leading dimension too large");
    const char* var = getenv("VIPANT_GEMM_VARIANT");       // read per call, as in vipant_gemm_nt
    const int fp8_dbg = var ? atoi(var) : 0;
    GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, bias, aux, lda, ldb, ldc, (int)M, (int)N, (int)K, 1.0f, fp8_dbg & (4194304 | (1 << 27) | (1 << 28)), sa, sb, 0, nullptr};
    p.cq = cq; p.cqs = cq_scale;
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case VIPANT_EPI_BF16:
            // the deep look-ahead schedule (VAR 10): -2..-4 % at K >= 3072, neutral at K <= 1024; bit 15 of VIPANT_GEMM_VARIANT: off
            // wide outputs (>= 8 column tiles, an even number): DEEP on the column-grouped walk -- the ViT-L qkv launch (N = 3072, K = 1024)
            // 1172 -> 1133 us, bit-identical; at N = 1024 (four column tiles) it gains nothing or loses.  Bit 26: off.
            if (!(fp8_dbg & (67108864 | 32768)) && ceil_div(M, BM) * ceil_div(N, BN) >= 256 && ceil_div(N, BN) % 2 == 0 && ceil_div(N, BN) >= 8)
                return launch_pp_variant<VIPANT_EPI_BF16, 12, 1>(p, s);
            if (!(fp8_dbg & 32768)) return launch_pp_variant<VIPANT_EPI_BF16, 10, 1>(p, s);
            return launch_pp_variant<VIPANT_EPI_BF16, 0, 1>(p, s);
        case VIPANT_EPI_QUICKGELU_D8:
        case VIPANT_EPI_DQUICKGELU_D8:
            VIPANT_REQUIRE((aux != nullptr || (C == nullptr && cq != nullptr)) && (uintptr_t)aux % 16 == 0, VIPANT_EBADSHAPE,
                           "gemm_nt_e4m3: the 8-bit QuickGELU' epilogues need a 16-byte aligned aux (the code matrix)");
            if (cq != nullptr && epilogue == VIPANT_EPI_QUICKGELU_D8 && C != nullptr && aux == nullptr) {
                vipant_set_error("gemm_nt_e4m3: C without aux (the code matrix)");
                return VIPANT_EBADSHAPE;
            }
            if (cq != nullptr && epilogue == VIPANT_EPI_QUICKGELU_D8 && C == nullptr && aux == nullptr)
                return launch_pp_variant<VIPANT_EPI_QUICKGELU_D8, 0, 1, 2>(p, s);           // the e4m3 form alone
            if (cq != nullptr && epilogue == VIPANT_EPI_QUICKGELU_D8 && C == nullptr) {      // the e4m3 form + the codes
                if (ceil_div(M, BM) * ceil_div(N, BN) >= 256 && ceil_div(N, BN) % 2 == 0 && !(fp8_dbg & 33554432))
                    return launch_pp_variant<VIPANT_EPI_QUICKGELU_D8, 8, 1, 3>(p, s);
                return launch_pp_variant<VIPANT_EPI_QUICKGELU_D8, 0, 1, 3>(p, s);
            }
            if (cq != nullptr && epilogue == VIPANT_EPI_DQUICKGELU_D8 && C == nullptr) {     // QuickGELU': the e4m3 form alone
                if (ceil_div(M, BM) * ceil_div(N, BN) >= 256 && ceil_div(N, BN) % 2 == 0 && !(fp8_dbg & 33554432))
                    return launch_pp_variant<VIPANT_EPI_DQUICKGELU_D8, 8, 1, 2>(p, s);
                return launch_pp_variant<VIPANT_EPI_DQUICKGELU_D8, 0, 1, 2>(p, s);
            }
            if (cq != nullptr) {
                // the column-grouped walk (k-step schedule; half of the weight bytes per XCD): 2186 -> 2149 us and 2329 -> 2305 us at the
                // ViT-L shape, bit-identical; DEEP on the grouped walk gains nothing here.  Bit 25 of VIPANT_GEMM_VARIANT: the plain walk.
                const bool groupable = ceil_div(M, BM) * ceil_div(N, BN) >= 256 && ceil_div(N, BN) % 2 == 0;
                if (groupable && !(fp8_dbg & 33554432))
                    return epilogue == VIPANT_EPI_QUICKGELU_D8 ? launch_pp_variant<VIPANT_EPI_QUICKGELU_D8, 8, 1, 1>(p, s)
                                                                : launch_pp_variant<VIPANT_EPI_DQUICKGELU_D8, 8, 1, 1>(p, s);
                return epilogue == VIPANT_EPI_QUICKGELU_D8 ? launch_pp_variant<VIPANT_EPI_QUICKGELU_D8, 0, 1, 1>(p, s)
                                                            : launch_pp_variant<VIPANT_EPI_DQUICKGELU_D8, 0, 1, 1>(p, s);
            }
            return epilogue == VIPANT_EPI_QUICKGELU_D8 ? launch_pp_variant<VIPANT_EPI_QUICKGELU_D8, 0, 1>(p, s)
                                                        : launch_pp_variant<VIPANT_EPI_DQUICKGELU_D8, 0, 1>(p, s);
        default:
            vipant_set_error("gemm_nt_e4m3: epilogue %d is not built for e4m3 operands (bf16 and the two 8-bit QuickGELU' ones are)", epilogue);
            return VIPANT_EBADSHAPE;
    }
}
