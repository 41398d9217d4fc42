// LDS-DMA throughput probe (MI355X): how fast can one CU pull data L2/HBM -> LDS with buffer_load ... lds,
// as a function of bytes in flight and of the number of issuing waves?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define LDS_AS __attribute__((address_space(3)))

// Each workgroup (NW waves) repeatedly fills `batch` KiB of LDS (wave w issues its share back to back), then waits
// vmcnt(0) + barrier.  mode 0: every workgroup streams its own region (HBM / L2 miss); mode 1: all read the same 64 KiB.
template <int NW>
__global__ __launch_bounds__(NW * 64) void probe(const char* src, size_t region, int batch_kb, int iters, int mode,
                                                 unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = mode ? src : src + (size_t)blockIdx.x * region;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(mode ? 65536 : region), 0x00020000);
    const int per_wave = batch_kb / NW;            // KiB (= DMA instructions) per wave per batch
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t off = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < per_wave; ++i) {
            const uint32_t o = mode ? ((wave * per_wave + i) * 1024) % 65536 : off + (wave * per_wave + i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + ((wave * per_wave + i) * 1024) % 131072), 16,
                                                     o + lane * 16, 0, 0, 0);
        }
        off += batch_kb * 1024;
        if (off + batch_kb * 1024 > region) off = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t region = 8u << 20;               // 8 MiB per workgroup -> 2 GiB total, far beyond L2 / MALL
    const int nwg = 256;
    char* src; unsigned long long* cyc;
    hipMalloc(&src, region * nwg); hipMemset(src, 1, region * nwg);
    hipMalloc(&cyc, nwg * 8);
    hipFuncSetAttribute((const void*)probe<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)probe<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int nw : {4, 8})
            for (int batch : {16, 32, 64, 128}) {
                const int iters = 512;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (nw == 8) hipLaunchKernelGGL(probe<8>, dim3(nwg), dim3(512), 131072, 0, src, region, batch, iters, mode, cyc);
                    else hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(256), 131072, 0, src, region, batch, iters, mode, cyc);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)nwg * iters * batch * 1024;
                printf("mode %s waves %d batch %3d KiB in flight: %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s chip\n",
                       mode ? "same-64KiB(L2 hit)" : "stream(HBM)       ", nw, batch, ms, bytes / nwg / ms / 1e6, bytes / ms / 1e9);
            }
    return 0;
}
