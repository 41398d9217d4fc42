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
            const int64_t o = (int64_t)m * p.ldc + n4;
            f32x4 v = acc[i][j] + bv;
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
                for (int e = 0; e < 4; ++e) g[e] = v[e] * fast_sigmoid(1.702f * v[e]);
                *(bf16x4*)((bf16_t*)p.C + o) = f32x4_to_bf16x4(g);
            } else if (EPI == VIPANT_EPI_DQUICKGELU) {
                const bf16x4 u4 = *(const bf16x4*)((const bf16_t*)p.aux + o);
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = (float)u4[e];
                    const float sg = fast_sigmoid(1.702f * u);
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
        bv[j] = (EPI != VIPANT_EPI_DQUICKGELU && p.bias != nullptr && n4 < p.N) ? *(const f32x4*)(p.bias + n4)
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
                            g[e] = (bf16_t)(u * fast_sigmoid(1.702f * u));
                        }
                        *(bf16x8*)((bf16_t*)p.C + o) = g;
                    } else {  // VIPANT_EPI_DQUICKGELU
                        const bf16x8 u8 = *(const bf16x8*)((const bf16_t*)p.aux + o);
                        bf16x8 d;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float u = (float)u8[e];
                            const float sg = fast_sigmoid(1.702f * u);
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

template <int EPI>
int32_t launch_persistent(const GemmNT& p, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_persistent_kernel<EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        configured = true;
    }
    const int64_t tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    int64_t grid = tiles < 256 ? (tiles + 7) / 8 * 8 : 256;
    hipLaunchKernelGGL(gemm_nt_persistent_kernel<EPI>, dim3((unsigned)grid), dim3(512), 2 * STAGE_BYTES, stream, p);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

template <int EPI>
int32_t launch(const GemmNT& p, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_kernel<EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES));
        configured = true;
    }
    const int64_t tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES, stream, p);
    VIPANT_LAUNCH_CHECK();
    return VIPANT_OK;
}

}  // namespace

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
    static const int dbg = getenv("VIPANT_GEMM_VARIANT") ? atoi(getenv("VIPANT_GEMM_VARIANT")) : 0;
    GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, bias, aux, lda, ldb, ldc, (int)M, (int)N, (int)K, alpha, dbg};
    hipStream_t s = (hipStream_t)stream;
    const bool staged = (N % 8 == 0) && (ldc % 8 == 0) && !(dbg & 2);
    switch (epilogue) {
        case VIPANT_EPI_BF16:
            return staged ? launch_persistent<VIPANT_EPI_BF16>(p, s) : launch<VIPANT_EPI_BF16>(p, s);
        case VIPANT_EPI_F32: return launch<VIPANT_EPI_F32>(p, s);
        case VIPANT_EPI_RESIDUAL_F32:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: residual epilogue needs aux");
            return staged ? launch_persistent<VIPANT_EPI_RESIDUAL_F32>(p, s) : launch<VIPANT_EPI_RESIDUAL_F32>(p, s);
        case VIPANT_EPI_QUICKGELU:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: quickgelu epilogue needs aux (U out)");
            return staged ? launch_persistent<VIPANT_EPI_QUICKGELU>(p, s) : launch<VIPANT_EPI_QUICKGELU>(p, s);
        case VIPANT_EPI_DQUICKGELU:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: dquickgelu epilogue needs aux (U in)");
            return staged ? launch_persistent<VIPANT_EPI_DQUICKGELU>(p, s) : launch<VIPANT_EPI_DQUICKGELU>(p, s);
        case VIPANT_EPI_SCALE_F32: return launch<VIPANT_EPI_SCALE_F32>(p, s);
        default:
            vipant_set_error("gemm_nt: unknown epilogue %d", epilogue);
            return VIPANT_EBADSHAPE;
    }
}
