// Retrieval evaluation (LossHead.report / retrieval_eval, cvap/module/decoder/loss_head.py:71-168): the reference
// forms sim = x1 x2^T, argsorts every row and looks up where the gold column landed.  The position of a column in a
// descending sort is the number of entries greater than it, so
//   ranks[i, g] = #{ j : sim[i, j] > sim[i, gold[i, g]] },      top1[i] = argmax_j sim[i, j]
// need neither the sort nor the N1 x N2 matrix in HBM (1 audio x 5 captions at N1 = 20k: 8 GB of fp32 + the int64
// argsort in the reference).
//
// gfx950 design: 256x256 similarity tiles on MFMA through the shared NT main loop with the same hi/lo bf16 split as
// the InfoNCE kernels (fp32-grade similarities: ordering differs from an fp32 GEMM only for gaps below ~1e-6).
//   pass 1  every tile publishes the similarity of the gold columns it contains and folds its per-row maximum into
//           a packed (ordered similarity, ~column) 64-bit atomicMax;
//   pass 2  recomputes the tiles (bit-identical arithmetic, so a gold entry never out-ranks itself) and counts, per
//           row and gold, the entries above the gold similarity: integer atomicAdd, order independent.
#include "nt_core.h"

namespace {

using namespace ntcore;

struct RetWs {
    bf16_t *x1cat, *x2cat;
    float* goldsim;
    unsigned long long* best;
    size_t total;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

RetWs carve(char* base, int64_t N1, int64_t N2, int64_t E, int64_t G) {
    RetWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base + off; off += align256(bytes); return p; };
    w.x1cat = (bf16_t*)take((size_t)N1 * 3 * E * 2);
    w.x2cat = (bf16_t*)take((size_t)N2 * 3 * E * 2);
    w.goldsim = (float*)take((size_t)N1 * G * 4);
    w.best = (unsigned long long*)take((size_t)N1 * 8);
    w.total = off;
    return w;
}

// x -> [hi | hi | lo] (queries, second = 0) or [hi | lo | hi] (candidates, second = 1)
__global__ __launch_bounds__(256) void ret_prep_kernel(const float* __restrict__ x, bf16_t* __restrict__ cat, int64_t total,
                                                       int E, int second) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / E;
        const int c = (int)(i % E);
        const float a = x[i];
        const bf16_t hi = (bf16_t)a, lo = (bf16_t)(a - (float)hi);
        bf16_t* p = cat + r * 3 * E + c;
        p[0] = hi; p[E] = second ? lo : hi; p[2 * E] = second ? hi : lo;
    }
}

__global__ __launch_bounds__(256) void ret_init_kernel(int32_t* ranks, float* goldsim, int64_t nranks, unsigned long long* best,
                                                       int64_t N1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nranks) { ranks[i] = 0; goldsim[i] = __builtin_nanf(""); }   // a gold index outside [0, N2) ranks 0
    if (i < N1) best[i] = 0ull;
}

__device__ __forceinline__ unsigned ordered_bits(float v) {     // monotone float -> unsigned
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int PASS>
__global__ __launch_bounds__(512, 2) void ret_tile_kernel(RetWs w, const int32_t* __restrict__ gold, int32_t* __restrict__ ranks,
                                                          int N1, int N2, int K, int G) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fq = lane >> 4;
    const int ntm = (N1 + BM - 1) / BM, ntn = (N2 + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, ntm * ntn);
    const int tm = tile / ntn, tn = tile % ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    f32x4 acc[8][4];
    mainloop(smem, w.x1cat, K, N1, w.x2cat, K, N2, K, m0, n0, wave, lane, acc);
    // lane holds sim[mb + 16 i][nb + 16 j + r]
    const int mb = m0 + wm * 128 + frow, nb = n0 + wn * 64 + fq * 4;

    if (PASS == 1) {
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mb + i * 16;
                if (m >= N1) continue;
                const int c = gold[(int64_t)m * G + g] - nb;          // column of the gold entry relative to this lane
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c == j * 16 + r) w.goldsim[(int64_t)m * G + g] = acc[i][j][r];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned long long key = 0ull;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nb + j * 16 + r;
                    if (n < N2) {
                        const unsigned long long k = ((unsigned long long)ordered_bits(acc[i][j][r]) << 32) | (unsigned)(~n);
                        key = k > key ? k : key;
                    }
                }
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const unsigned long long other = __shfl_xor(key, o, 64);
                key = other > key ? other : key;
            }
            const int m = mb + i * 16;
            if (fq == 0 && m < N1 && key != 0ull) atomicMax(w.best + m, key);
        }
    } else {
        for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mb + i * 16;
                const float gs = m < N1 ? w.goldsim[(int64_t)m * G + g] : 0.f;
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cnt += (nb + j * 16 + r < N2 && acc[i][j][r] > gs) ? 1 : 0;
                cnt += __shfl_xor(cnt, 16, 64);
                cnt += __shfl_xor(cnt, 32, 64);
                if (fq == 0 && m < N1 && cnt != 0) atomicAdd(ranks + (int64_t)m * G + g, cnt);
            }
        }
    }
}

__global__ __launch_bounds__(256) void ret_top1_kernel(const unsigned long long* best, int32_t* top1, int64_t N1) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N1) top1[i] = (int32_t)(~(unsigned)(best[i] & 0xFFFFFFFFull));
}

}  // namespace

extern "C" size_t vipant_retrieval_workspace_bytes(int64_t N1, int64_t N2, int64_t E, int64_t G) {
    return carve(nullptr, N1, N2, E, G).total;
}

extern "C" int32_t vipant_retrieval_ranks(const float* x1, const float* x2, const int32_t* gold, int32_t* ranks, int32_t* top1,
                                          int64_t N1, int64_t N2, int64_t E, int64_t G, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    VIPANT_REQUIRE(N1 > 0 && N2 > 0 && E > 0 && E % 64 == 0, VIPANT_EBADSHAPE, "retrieval: need E %% 64 == 0 (N1=%ld N2=%ld E=%ld)",
                   (long)N1, (long)N2, (long)E);
    VIPANT_REQUIRE(G >= 1 && G <= 64 && gold != nullptr && ranks != nullptr, VIPANT_EBADSHAPE,
                   "retrieval: need 1 <= G <= 64 gold columns per query (G=%ld)", (long)G);
    VIPANT_REQUIRE(N1 * G < (1ll << 31) && N2 < (1ll << 31) && 256 * 3 * E * 2 < (1ll << 31), VIPANT_EBADSHAPE,
                   "retrieval: problem too large");
    VIPANT_REQUIRE(workspace != nullptr && workspace_bytes >= vipant_retrieval_workspace_bytes(N1, N2, E, G), VIPANT_ENOWORKSPACE,
                   "retrieval: workspace too small");
    VIPANT_REQUIRE((uintptr_t)workspace % 256 == 0, VIPANT_EALIGN, "retrieval: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const RetWs w = carve((char*)workspace, N1, N2, E, G);
    static DeviceOnce once;
    if (first_on_device(once)) {
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)ret_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        VIPANT_HIP_TRY(hipFuncSetAttribute((const void*)ret_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        done_on_device(once);
    }
    auto blocks = [](int64_t n) { const int64_t b = ceil_div(n, 256); return (unsigned)(b > 4096 ? 4096 : b); };
    hipLaunchKernelGGL(ret_prep_kernel, dim3(blocks(N1 * E)), dim3(256), 0, s, x1, w.x1cat, N1 * E, (int)E, 0);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ret_prep_kernel, dim3(blocks(N2 * E)), dim3(256), 0, s, x2, w.x2cat, N2 * E, (int)E, 1);
    VIPANT_LAUNCH_CHECK();
    const int64_t ninit = N1 * G;
    hipLaunchKernelGGL(ret_init_kernel, dim3((unsigned)ceil_div(ninit, 256)), dim3(256), 0, s, ranks, w.goldsim, ninit, w.best, N1);
    VIPANT_LAUNCH_CHECK();
    const unsigned tiles = (unsigned)(ceil_div(N1, BM) * ceil_div(N2, BN));
    const int K = (int)(3 * E);
    hipLaunchKernelGGL(ret_tile_kernel<1>, dim3(tiles), dim3(512), LDS_BYTES, s, w, gold, ranks, (int)N1, (int)N2, K, (int)G);
    VIPANT_LAUNCH_CHECK();
    hipLaunchKernelGGL(ret_tile_kernel<2>, dim3(tiles), dim3(512), LDS_BYTES, s, w, gold, ranks, (int)N1, (int)N2, K, (int)G);
    VIPANT_LAUNCH_CHECK();
    if (top1 != nullptr) {
        hipLaunchKernelGGL(ret_top1_kernel, dim3((unsigned)ceil_div(N1, 256)), dim3(256), 0, s, w.best, top1, N1);
        VIPANT_LAUNCH_CHECK();
    }
    return VIPANT_OK;
}
