// Probe for the next NT-contraction kernel (profiles/r4_gemm_experiments.md section 7): four-wave workgroups with a 128 x 256 tile,
// two per CU and independent (so one's store drain hides behind the other's main loop); A rows through LDS (LDS-DMA, two 16-KiB
// stages), each wave's 64 columns of B as MFMA fragments straight from global memory / L2 (no LDS, no DMA issue for B).
// Plain bf16 epilogue straight from the accumulator layout.  Timing only + a spot check; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -o gemm_w4_probe gemm_w4_probe.hip ; ./gemm_w4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

struct P { const bf16_t* A; const bf16_t* B; bf16_t* C; int M, N, K; int stores; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_base, 16, voff, soff, 0, 0);
}

__global__ __launch_bounds__(256, 2) void w4_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 stages x [128 rows][128 B]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int ntn = p.N / 256;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * 128, n0 = tn * 256 + 64 * wave;
    const int64_t abytes = ((int64_t)(p.M - m0) * p.K) * 2;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A + (int64_t)m0 * p.K, (uint32_t)(abytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : abytes));
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = (uint32_t)(row * p.K * 2 + c * 16);
    }
    auto fill = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_dma16(ra, smem + stage * 16384 + (wave * 4 + i) * 1024, voff[i], (uint32_t)kt * 128);
    };
    const int fs = (r >> 1) & 7;
    uint32_t offA[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offA[ks] = (uint32_t)(r * 128 + (((ks * 4 + g) ^ fs) << 4));
    const bf16_t* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = p.B + (int64_t)(n0 + 16 * j + r) * p.K + 8 * g;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = p.K / 64;
    bf16x8 bc[2][4], bn[2][4];
    fill(0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) bc[ks][j] = *(const bf16x8*)(bp[j] + 32 * ks);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            fill((kt + 1) & 1, kt + 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) bn[ks][j] = *(const bf16x8*)(bp[j] + 64 * (kt + 1) + 32 * ks);
        }
        const char* sa = smem + (kt & 1) * 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16x8 a = *(const bf16x8*)(sa + i * 2048 + offA[ks]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[ks][j], a, acc[i][j], 0, 0, 0);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) bc[ks][j] = bn[ks][j];
    }
    if (p.stores) {
        // staged: two rounds of 64 rows x 512 B through the (now idle) A stages, 16-byte chunk index XOR (row & 15); out as whole rows
        const int nt0 = tn * 256;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = rd * 4 + ii, row = ii * 16 + r;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16_t)acc[i][j][e];
                    const int colbyte = wave * 128 + j * 32 + g * 8;
                    *(bf16x4*)(smem + row * 512 + (((colbyte >> 4) ^ (row & 15)) << 4) + (colbyte & 8)) = o;
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int idx = t * 256 + threadIdx.x;
                const int R = idx >> 5, ch = idx & 31;
                const int m = m0 + rd * 64 + R;
                if (m < p.M) *(f32x4*)(p.C + (int64_t)m * p.N + nt0 + ch * 8) = *(const f32x4*)(smem + R * 512 + ((ch ^ (R & 15)) << 4));
            }
            __syncthreads();
        }
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int shapes[3][3] = {{161792, 3072, 768}, {161792, 2304, 768}, {161792, 768, 3072}};
    hipFuncSetAttribute((const void*)w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K);
        srand(7);
        for (auto& v : ha) v = f2bf((float)((rand() % 7) - 3) * 0.25f);
        for (auto& v : hb) v = f2bf((float)((rand() % 5) - 2) * 0.125f);
        bf16_t *da, *db, *dc;
        hipMalloc(&da, ha.size() * 2); hipMalloc(&db, hb.size() * 2); hipMalloc(&dc, (size_t)M * N * 2);
        hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        const int grid = (M / 128) * (N / 256);
        for (int stores = 1; stores >= 0; --stores) {
            P p{da, db, dc, M, N, K, stores};
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(w4_kernel, dim3(grid), dim3(256), 32768, 0, p);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(w4_kernel, dim3(grid), dim3(256), 32768, 0, p);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("M=%d N=%d K=%d stores=%d: %.1f us  (%.2f PFLOP/s)\n", M, N, K, stores, ms * 100.f, 2.0 * M * N * K / (ms * 1e-4) / 1e15);
        }
        P p{da, db, dc, M, N, K, 1};
        hipLaunchKernelGGL(w4_kernel, dim3(grid), dim3(256), 32768, 0, p);
        hipDeviceSynchronize();
        double worst = 0;
        for (int t = 0; t < 64; ++t) {
            const int m = (int)((int64_t)rand() * 7919 % M), n = rand() % N;
            uint16_t hc; hipMemcpy(&hc, (uint16_t*)dc + (size_t)m * N + n, 2, hipMemcpyDeviceToHost);
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)bf2f(ha[(size_t)m * K + k]) * bf2f(hb[(size_t)n * K + k]);
            const double err = fabs(bf2f(hc) - ref) / (fabs(ref) + 1.0);
            if (err > worst) worst = err;
        }
        printf("   spot check: worst relative error %.3e\n", worst);
        hipFree(da); hipFree(db); hipFree(dc);
    }
    return 0;
}
