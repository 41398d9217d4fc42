// C[M,N] = A[M,K] . B[N,K]^T with fused epilogues -- the dense contraction behind every nn.Linear /
// conv-as-GEMM on the path (cvap/module/val.py:245-247, 500-506; forward and the dX backward).
//
// gfx950 design: 256x256 output tile per 512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave),
// BK = 64, two 64 KiB LDS stages filled by LDS-DMA (buffer_load ... lds, 16 B per lane, out-of-range
// rows read as zero through the buffer descriptor), XOR-swizzled 128-B rows so that ds_read_b128
// fragment reads are bank-conflict free (swizzle applied on the DMA source address and on the read
// address: the LDS image of a DMA is lane-linear), v_mfma_f32_16x16x32_bf16 with the operands swapped
// (D = B_tile . A_tile^T) so that every lane ends up holding 4 consecutive output columns of one row.
#include "common.h"

#include "nt_core.h"

namespace {

using namespace ntcore;

struct GemmNT {
    const bf16_t* A; const bf16_t* B; void* C; const float* bias; void* aux;
    int64_t lda, ldb, ldc;
    int M, N, K;
    float alpha;
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
        if (n4 >= p.N) continue;
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
    GemmNT p{(const bf16_t*)A, (const bf16_t*)B, C, bias, aux, lda, ldb, ldc, (int)M, (int)N, (int)K, alpha};
    hipStream_t s = (hipStream_t)stream;
    switch (epilogue) {
        case VIPANT_EPI_BF16: return launch<VIPANT_EPI_BF16>(p, s);
        case VIPANT_EPI_F32: return launch<VIPANT_EPI_F32>(p, s);
        case VIPANT_EPI_RESIDUAL_F32:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: residual epilogue needs aux");
            return launch<VIPANT_EPI_RESIDUAL_F32>(p, s);
        case VIPANT_EPI_QUICKGELU:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: quickgelu epilogue needs aux (U out)");
            return launch<VIPANT_EPI_QUICKGELU>(p, s);
        case VIPANT_EPI_DQUICKGELU:
            VIPANT_REQUIRE(aux != nullptr, VIPANT_EBADSHAPE, "gemm_nt: dquickgelu epilogue needs aux (U in)");
            return launch<VIPANT_EPI_DQUICKGELU>(p, s);
        case VIPANT_EPI_SCALE_F32: return launch<VIPANT_EPI_SCALE_F32>(p, s);
        default:
            vipant_set_error("gemm_nt: unknown epilogue %d", epilogue);
            return VIPANT_EBADSHAPE;
    }
}
