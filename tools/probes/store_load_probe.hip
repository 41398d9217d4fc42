// Can a CU stream stores towards HBM (the GEMM tile epilogue) and pull LDS-DMA loads (the next tile's main loop) at the
// same time without either slowing down?  8 waves per workgroup, one workgroup per CU: waves 0-3 issue
// buffer_load ... lds (16 B / lane, batches of 32 KiB, vmcnt(0) between batches), waves 4-7 issue 16-B global stores of
// whole 512-B rows.  Reports per-CU rates for loads alone, stores alone and both together.
// Build: hipcc --offload-arch=gfx950 -O3 store_load_probe.hip -o store_load_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_AS __attribute__((address_space(3)))

__global__ __launch_bounds__(512) void probe(const char* src, char* dst, size_t region, int iters, int do_load, int do_store,
                                             int l2hit, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (do_load) {
            const char* base = l2hit ? src : src + (size_t)blockIdx.x * region;
            auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (uint32_t)(l2hit ? 65536 : region), 0x00020000);
            uint32_t off = 0;
            for (int it = 0; it < iters; ++it) {
                for (int i = 0; i < 8; ++i) {                        // 4 waves x 8 KiB = 32 KiB per batch
                    const uint32_t o = l2hit ? ((wave * 8 + i) * 1024) : off + (wave * 8 + i) * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + (wave * 8 + i) * 1024), 16, o + lane * 16, 0, 0, 0);
                }
                off += 32768;
                if (off + 32768 > region) off = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    } else if (do_store) {
        char* base = dst + (size_t)blockIdx.x * region;
        uint4 v = make_uint4(lane, wave, 3, 4);
        uint32_t off = 0;
        for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < 8; ++i)                              // 4 waves x 8 KiB = 32 KiB per batch
                *(uint4*)(base + off + ((wave - 4) * 8 + i) * 1024 + lane * 16) = v;
            off += 32768;
            if (off + 32768 > region) off = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const size_t region = 8u << 20;
    const int nwg = 256;
    char *src, *dst; unsigned long long* cyc;
    hipMalloc(&src, region * nwg); hipMemset(src, 1, region * nwg);
    hipMalloc(&dst, region * nwg);
    hipMalloc(&cyc, nwg * 8 * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1024;
    const double bytes = (double)iters * 32768;
    for (int l2hit = 0; l2hit < 2; ++l2hit)
        for (int mode = 1; mode <= 3; ++mode) {
            const int dl = mode & 1, ds = (mode >> 1) & 1;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 65536, 0, src, dst, region, iters, dl, ds, l2hit, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            unsigned long long h[nwg * 8];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double lc = 0, sc = 0;                                    // mean cycles (100 MHz memtime ticks) of loader / storer waves
            for (int b = 0; b < nwg; ++b) { lc += h[b * 8 + 0]; sc += h[b * 8 + 4]; }
            lc /= nwg; sc /= nwg;
            printf("loads %-8s %s%s: kernel %7.3f ms", l2hit ? "L2-hit" : "stream", dl ? "load " : "     ", ds ? "store" : "     ", ms);
            if (dl) printf("  load %6.1f GB/s/CU", bytes / (lc * 10.0));     // memtime ticks are 10 ns
            if (ds) printf("  store %6.1f GB/s/CU", bytes / (sc * 10.0));
            printf("\n");
        }
    return 0;
}
