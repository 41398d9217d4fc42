// 256x256x64 NT main loop shared by the GEMM and the fused InfoNCE tile kernels (see gemm_nt.hip for the design).
#pragma once
#include "common.h"

namespace ntcore {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 64 KiB
constexpr int A_BYTES = BM * BK * 2;
constexpr int LDS_BYTES = 2 * STAGE_BYTES;

// acc[i][j][r] = C[m0 + wm*128 + i*16 + (lane&15)][n0 + wn*64 + j*16 + (lane>>4)*4 + r]
// for A[M,K] (rows m0..), B[N,K] (rows n0..), both K-contiguous bf16; rows >= M / N contribute zeros.
__device__ __forceinline__ void mainloop(char* smem, const bf16_t* A, int64_t lda, int M, const bf16_t* B, int64_t ldb,
                                         int N, int K, int m0, int n0, int wave, int lane, f32x4 (&acc)[8][4]) {
    const int wm = wave >> 2, wn = wave & 3;
    struct { const bf16_t* A; const bf16_t* B; int64_t lda, ldb; int M, N, K; } p{A, B, lda, ldb, M, N, K};
    // Buffer descriptors based at the tile's first row: rows >= M (N) fall outside and read as zero.
    const bf16_t* Ab = p.A + (int64_t)m0 * p.lda;
    const bf16_t* Bb = p.B + (int64_t)n0 * p.ldb;
    const int64_t a_bytes = ((int64_t)(p.M - m0) * p.lda - (p.lda - p.K)) * 2;
    const int64_t b_bytes = ((int64_t)(p.N - n0) * p.ldb - (p.ldb - p.K)) * 2;
    const auto rsA = make_rsrc(Ab, (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes));
    const auto rsB = make_rsrc(Bb, (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes));

    // DMA source offsets: wave w fills row blocks w*4 .. w*4+3 (8 rows x 128 B each) of A and of B.
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffA[i] = (uint32_t)(r * p.lda * 2 + c * 16);
        voffB[i] = (uint32_t)(r * p.ldb * 2 + c * 16);
    }
    auto stage_load = [&](int stage, int kt) {
        char* sA = smem + stage * STAGE_BYTES + wave * 4096;
        char* sB = sA + A_BYTES;
        const uint32_t koff = (uint32_t)kt * (BK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, sA + i * 1024, voffA[i], koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsB, sB + i * 1024, voffB[i], koff);
    };

    // Fragment read offsets (bytes) for the two 32-deep k-steps of a stage.
    const int frow = lane & 15, fq = lane >> 4, fs = (lane >> 1) & 7;
    uint32_t offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const uint32_t cb = (uint32_t)(((ks * 4 + fq) ^ fs) << 4);
        offA[ks] = (uint32_t)((wm * 128 + frow) * 128) + cb;
        offB[ks] = (uint32_t)(A_BYTES + (wn * 64 + frow) * 128) + cb;
    }

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage_load(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage_load(cur ^ 1, kt + 1);
        const char* s = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[8], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(s + offB[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(s + offA[ks] + i * 2048);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

}

}  // namespace ntcore
