// Throughput of ds_read_b64_tr_b8 against ds_read_b64_tr_b16 and ds_read_b64: 8 waves per CU (one 512-thread workgroup per CU, as the
// contraction kernels run), each wave issuing 64 reads per iteration on a conflict-free image, no waits inside the iteration.
//   hipcc --offload-arch=gfx950 -O2 tr_rate_probe.hip -o tr_rate_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(512) void probe(uint32_t* out, int iters, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((uint32_t*)smem)[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // conflict-free per 32-lane half: lane i reads 8 bytes at (wave * 8 KiB) + i * 8  (linear)
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + wave * 8192 + lane * 8;
    int __attribute__((ext_vector_type(2))) acc = {0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int __attribute__((ext_vector_type(2))) v;
            if (KIND == 0) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
            if (KIND == 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
            if (KIND == 2) asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(8)");
            acc[0] ^= v[0]; acc[1] ^= v[1];
            addr ^= 512u * (k & 7);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] ^ acc[1];
}

template <int KIND>
void run(const char* name) {
    uint32_t* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void*)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 2000;
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 65536, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 65536, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double reads = 8.0 * 16 * iters;                       // wave-instructions per CU
    printf("%-22s %.3f ms; %.2f cycles per wave-instruction per CU (clock64), %.1f B/clk/CU\n", name, ms, (double)h[0] / reads, 512.0 * reads / (double)h[0]);
}

int main() {
    run<0>("ds_read_b64");
    run<1>("ds_read_b64_tr_b16");
    run<2>("ds_read_b64_tr_b8");
    return 0;
}
