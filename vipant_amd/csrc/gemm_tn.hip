// C[P,Q] (+)= A[M,P]^T . B[M,Q] -- the weight-gradient contraction (autograd of nn.Linear / conv1 weights,
// cvap/module/val.py:500-506, 245-247), reduction over the token dimension M.
//
// gfx950 design: both operands are M-major (the reduction index is the slow one), so tiles are staged
// [64 m][256 cols] by LDS-DMA exactly as they lie in HBM (full 512-B row segments, coalesced) and the MFMA
// fragments -- which need 8 consecutive reduction elements per lane -- are formed by ds_read_b64_tr_b16
// (hardware transpose read), two per 16x16x32 operand.  512-B rows are XOR-swizzled at 16-B granularity
// (on the DMA source address and on the read address) so the 32 lanes of a half-wave hit all 64 banks.
// 256x256 output tile per 512-thread workgroup; the M range is split over workgroups to fill 256 CUs and
// the fp32 partial tiles are summed by a second deterministic pass (no float atomics).
#include "common.h"

namespace {

constexpr int TP = 256, TQ = 256, BK = 64;
constexpr int OP_BYTES = BK * TP * 2;       // 32 KiB per operand tile
constexpr int STAGE_BYTES = 2 * OP_BYTES;   // 64 KiB

struct GemmTN {
    const bf16_t* A; const bf16_t* B; float* out;  // out: slab base (splits > 1) or C
    int64_t lda, ldb, ldo;
    int M, P, Q;
    int splits, kt_per_split;
    int direct;   // 1: write straight into C (single split, no accumulate)
    float* colsum;   // optional [splits * ntq][ntp*256]: partial column sums of A (bias gradient of the same Linear)
};

__device__ __forceinline__ int tn_swz(int m) { return ((m & 3) | (((m >> 3) & 1) << 2)) << 1; }

__global__ __launch_bounds__(512, 2) void gemm_tn_kernel(GemmTN p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wq = wave & 3;

    const int ntq = (p.Q + TQ - 1) / TQ, ntp = (p.P + TP - 1) / TP;
    const int ntiles = ntp * ntq;
    const int bid = xcd_remap(blockIdx.x, ntiles * p.splits);
    const int split = bid / ntiles, tile = bid % ntiles;
    const int tp = tile / ntq, tq = tile % ntq;
    const int p0 = tp * TP, q0 = tq * TQ;
    const int nk_total = (p.M + BK - 1) / BK;
    const int kt0 = split * p.kt_per_split;
    int nk = nk_total - kt0;
    if (nk > p.kt_per_split) nk = p.kt_per_split;
    const int mbeg = kt0 * BK;

    // Descriptors based at (first row of this split, first column of the tile).  Rows >= M are past the end
    // of the range and read as zero, which is what the reduction needs for the M tail.
    const bf16_t* Ab = p.A + (int64_t)mbeg * p.lda + p0;
    const bf16_t* Bb = p.B + (int64_t)mbeg * p.ldb + q0;
    int64_t a_bytes = ((int64_t)(p.M - mbeg) * p.lda - p0) * 2;
    int64_t b_bytes = ((int64_t)(p.M - mbeg) * p.ldb - q0) * 2;
    if (a_bytes < 0) a_bytes = 0;
    if (b_bytes < 0) b_bytes = 0;
    const auto rsA = make_rsrc(Ab, (uint32_t)(a_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_bytes));
    const auto rsB = make_rsrc(Bb, (uint32_t)(b_bytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : b_bytes));

    // DMA: one wave instruction = 2 tile rows (m) x 512 B.  Wave w fills rows w*8 .. w*8+7 of A and of B.
    uint32_t voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = (wave * 4 + i) * 2 + (lane >> 5);
        const int c = (lane & 31) ^ tn_swz(m);
        voffA[i] = (uint32_t)(m * p.lda * 2 + c * 16);
        voffB[i] = (uint32_t)(m * p.ldb * 2 + c * 16);
    }
    const uint32_t kstepA = (uint32_t)(BK * p.lda * 2), kstepB = (uint32_t)(BK * p.ldb * 2);
    auto stage_load = [&](int stage, int kt) {
        char* sA = smem + stage * STAGE_BYTES + wave * 4096;
        char* sB = sA + OP_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsA, sA + i * 1024, voffA[i], (uint32_t)kt * kstepA);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(rsB, sB + i * 1024, voffB[i], (uint32_t)kt * kstepB);
    };

    // Transposed fragment reads.  Lane l: g = l>>4 owns k-slots 8g..8g+7 (tile rows ks*32 + 8g + 0..7),
    // within the group lane 4*qq+pp supplies row qq (first read) / 4+qq (second), columns 4pp..4pp+3 of the
    // 16-column block; it receives column (l&15), rows 0..3 / 4..7.
    const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
    uint32_t rdoff[2][2];   // [ks][half] byte offset of this lane's address for column block 0 of the tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ks * 32 + 8 * g + 4 * h + qq;
            rdoff[ks][h] = (uint32_t)(m * 512 + (((pp >> 1) ^ tn_swz(m)) << 4) + (pp & 1) * 8);
        }
    // Column block cb (16 columns = 2 chunks) adds (2*cb) to the logical chunk; since tn_swz only touches
    // bits 1..3 and 2*cb has bit 0 clear, (c0 + 2cb) ^ s == (c0 ^ s) ^ (2cb)  ->  XOR the byte offset with cb<<5.
    const uint32_t lds0 = lds_offset(smem);
    auto frag = [&](uint32_t tile, int ks, int cb) -> bf16x8 {      // raw (asm) reads: see lds_read_tr16_pair_raw in common.h
        return lds_read_tr16_pair_raw(tile + (rdoff[ks][0] ^ (uint32_t)(cb << 5)), tile + (rdoff[ks][1] ^ (uint32_t)(cb << 5)));
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Column sums of A over the token dimension ride along: the A tile is in LDS anyway.  Thread t owns column
    // (t & 255) and the rows of half (t >> 8) of every K-tile; only the workgroups of the first Q tile do it.
    // Column sums of A ride along.  The ntq workgroups that share an A tile take turns (K-tile kt belongs to workgroup
    // kt % ntq), and a thread sums 8 columns (one 16-B chunk) of 4 rows per K-tile with ds_read_b128 -- the first version
    // (the tq == 0 workgroups alone, 32 two-byte reads per thread and K-tile) made those workgroups, and with them the
    // whole launch, 15-25 % slower.
    const bool do_colsum = p.colsum != nullptr;
    const int cs_chunk = tid & 31, cs_rg = tid >> 5;
    float cs_acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs_acc[e] = 0.f;
    if (nk > 0) {
        stage_load(0, 0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage_load(cur ^ 1, kt + 1);
            const char* sA = smem + cur * STAGE_BYTES;
            const char* sB = sA + OP_BYTES;
            if (do_colsum && (kt % ntq) == tq) {
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int m = cs_rg + 16 * k4;
                    const bf16x8 v = *(const bf16x8*)(sA + m * 512 + ((cs_chunk ^ tn_swz(m)) << 4));
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs_acc[e] += (float)v[e];
                }
            }
            // Explicit fragment pipeline (same as the NT kernel): the B fragments of the whole K-tile first, the A
            // fragments in a 3-deep register ring two MFMA groups ahead of their use.  The transposed reads go through
            // inline asm (the builtin would make hipcc drain the LDS-DMA of the NEXT stage, issued just above, before
            // touching this one), so the lgkmcnt waits are explicit: LDS results return in order, a fragment = 2 reads.
            const uint32_t tA = lds0 + (uint32_t)(cur * STAGE_BYTES), tB = tA + OP_BYTES;
            bf16x8 fq[2][4], fp[3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) fq[ks][j] = frag(tB, ks, wq * 4 + j);
            fp[0] = frag(tA, 0, wp * 8 + 0);
            fp[1] = frag(tA, 0, wp * 8 + 1);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ks = t >> 3, i = t & 7;
                if (t + 2 < 16) fp[(t + 2) % 3] = frag(tA, (t + 2) >> 3, wp * 8 + ((t + 2) & 7));
                // outstanding behind fragment t: the fragments of groups t+1 and t+2 (2 reads each)
                if (t + 2 < 16) lds_raw_wait<4>(); else if (t + 1 < 16) lds_raw_wait<2>(); else lds_raw_wait<0>();
                lds_raw_use(fp[t % 3]);
                if (t == 0) {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) lds_raw_use(fq[k2][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[ks][j], fp[t % 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            cur ^= 1;
        }
    }

    if (do_colsum) {      // all LDS reads of the loop are behind its last barrier: reuse the front of the buffer
        float* red = (float*)smem;                     // [16 row groups][256 columns]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[cs_rg * 256 + cs_chunk * 8 + e] = cs_acc[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += red[r * 256 + tid];
            p.colsum[((int64_t)split * ntq + tq) * (ntp * TP) + p0 + tid] = sum;
        }
    }
    // lane holds C[p = p0 + wp*128 + i*16 + (lane&15)][q = q0 + wq*64 + j*16 + (lane>>4)*4 + 0..3]
    const int frow = lane & 15;
    if (p.direct) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pr = p0 + wp * 128 + i * 16 + frow;
            if (pr >= p.P) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qc = q0 + wq * 64 + j * 16 + g * 4;
                if (qc < p.Q) *(f32x4*)(p.out + (int64_t)pr * p.ldo + qc) = acc[i][j];
            }
        }
    } else {
        float* slab = p.out + ((int64_t)split * ntiles + tile) * (TP * TQ);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(f32x4*)(slab + (wp * 128 + i * 16 + frow) * TQ + wq * 64 + j * 16 + g * 4) = acc[i][j];
    }
}

__global__ void gemm_tn_reduce_kernel(const float* slab, float* C, int64_t ldc, int P, int Q, int splits,
                                      int accumulate) {
    const int ntq = (Q + TQ - 1) / TQ, ntp = (P + TP - 1) / TP;
    const int ntiles = ntp * ntq;
    const int64_t total4 = (int64_t)ntiles * TP * TQ / 4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = idx * 4;
        const int tile = (int)(e / (TP * TQ));
        const int r = (int)(e % (TP * TQ)) / TQ, c = (int)(e % TQ);
        const int pr = (tile / ntq) * TP + r, qc = (tile % ntq) * TQ + c;
        if (pr >= P || qc >= Q) continue;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < splits; ++k) s += *(const f32x4*)(slab + ((int64_t)k * ntiles + tile) * (TP * TQ) + r * TQ + c);
        float* dst = C + (int64_t)pr * ldc + qc;
        if (accumulate) s += *(const f32x4*)dst;
        *(f32x4*)dst = s;
    }
}

// out[i] (+)= sum over the `parts` partial vectors.  64 columns per 256-thread block, 4 threads per column each taking every
// fourth partial with independent loads in flight (a single thread walking 20-80 partials one after the other made this tiny
// kernel 20-90 us long).
__global__ __launch_bounds__(256) void gemm_tn_colsum_reduce_kernel(const float* part, float* out, int P, int stride, int parts,
                                                                    int accumulate) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + c;
    float s0 = 0.f, s1 = 0.f;
    if (i < P) {
        int k = sub;
        for (; k + 4 < parts; k += 8) { s0 += part[(int64_t)k * stride + i]; s1 += part[(int64_t)(k + 4) * stride + i]; }
        if (k < parts) s0 += part[(int64_t)k * stride + i];
    }
    red[sub][c] = s0 + s1;
    __syncthreads();
    if (sub == 0 && i < P) {
        const float s = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        out[i] = accumulate ? out[i] + s : s;
    }
}

void plan(int64_t M, int64_t P, int64_t Q, int* splits, int* kt_per_split) {
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    const int64_t nk = ceil_div(M, BK);
    int64_t s = 256 / tiles;
    if (s < 1) s = 1;
    if (s > nk) s = nk;
    const int64_t per = ceil_div(nk, s);
    *splits = (int)ceil_div(nk, per);
    *kt_per_split = (int)per;
}

}  // namespace

extern "C" size_t vipant_gemm_tn_workspace_bytes(int64_t M, int64_t P, int64_t Q) {
    int splits, per;
    plan(M, P, Q, &splits, &per);
    return (size_t)splits * (size_t)(ceil_div(P, TP) * ceil_div(Q, TQ)) * TP * TQ * sizeof(float) +
           (size_t)splits * (size_t)ceil_div(Q, TQ) * (size_t)ceil_div(P, TP) * TP * sizeof(float);
}

extern "C" int32_t vipant_gemm_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C,
                                  int64_t ldc, int64_t M, int64_t P, int64_t Q, int32_t accumulate, float* a_colsum,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(M > 0 && P > 0 && Q > 0, VIPANT_EBADSHAPE, "gemm_tn: empty problem");
    VIPANT_REQUIRE(Q % 4 == 0, VIPANT_EBADSHAPE, "gemm_tn: need Q%%4==0 (P=%ld Q=%ld)",
                   (long)P, (long)Q);
    VIPANT_REQUIRE(lda >= P && ldb >= Q && ldc >= Q && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, VIPANT_EALIGN,
                   "gemm_tn: bad leading dims");
    VIPANT_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0), VIPANT_EALIGN,
                   "gemm_tn: operands must be 16-byte aligned");
    int splits, per;
    plan(M, P, Q, &splits, &per);
    VIPANT_REQUIRE((int64_t)(per + 1) * BK * (lda > ldb ? lda : ldb) * 2 < (1ll << 32), VIPANT_EBADSHAPE,
                   "gemm_tn: per-split byte range exceeds 4 GiB");
    const size_t need = vipant_gemm_tn_workspace_bytes(M, P, Q);
    const int direct = (splits == 1 && !accumulate) ? 1 : 0;
    if (!direct || a_colsum != nullptr)
        VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= need, VIPANT_ENOWORKSPACE,
                       "gemm_tn: workspace too small (%zu < %zu)", workspace_bytes, need);
    static bool configured = false;
    if (!configured) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           2 * STAGE_BYTES));
        configured = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = ceil_div(P, TP) * ceil_div(Q, TQ);
    const size_t slab_bytes = (size_t)splits * (size_t)tiles * TP * TQ * sizeof(float);
    float* cs_part = a_colsum != nullptr ? (float*)((char*)workspace + slab_bytes) : nullptr;
    GemmTN p{(const bf16_t*)A, (const bf16_t*)B, direct ? C : (float*)workspace, lda, ldb, direct ? ldc : TQ,
             (int)M, (int)P, (int)Q, splits, per, direct, cs_part};
    hipLaunchKernelGGL(gemm_tn_kernel, dim3((unsigned)(tiles * splits)), dim3(512), 2 * STAGE_BYTES, s, p);
    VIPANT_LAUNCH_CHECK();
    if (!direct) {
        const int64_t total4 = tiles * TP * TQ / 4;
        int blocks = (int)ceil_div(total4, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, C, ldc,
                           (int)P, (int)Q, splits, accumulate);
        VIPANT_LAUNCH_CHECK();
    }
    if (a_colsum != nullptr) {
        hipLaunchKernelGGL(gemm_tn_colsum_reduce_kernel, dim3((unsigned)ceil_div(P, 64)), dim3(256), 0, s,
                           (const float*)cs_part, a_colsum, (int)P, (int)(ceil_div(P, TP) * TP), (int)(splits * ceil_div(Q, TQ)), accumulate);
        VIPANT_LAUNCH_CHECK();
    }
    return VIPANT_OK;
}
