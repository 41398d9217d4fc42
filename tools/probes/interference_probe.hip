// Does LDS-DMA issued by one wave slow the MFMA stream of its SIMD partner?  (MI355X micro-probe)
// 8 waves per workgroup, one workgroup per CU.  Waves 0-3 (one per SIMD) run a pure MFMA loop and time it with
// s_memtime; waves 4-7 (their SIMD partners) run `mode`:
//   0 exit at once   1 LDS-DMA loop (L2-hot 64 KiB source)   2 ds_read_b128 loop   3 DMA on wave 4 only (SIMD 0)
//   4 plain global_load_dwordx4 loop (to registers)            5 LDS-DMA + the MFMA waves ALSO read LDS fragments
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_AS __attribute__((address_space(3)))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void probe(const char* src, float* sink, unsigned long long* cyc, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 65536, 0x00020000);
    if (wave < 4) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(lane * 0.01f + e); b[e] = (__bf16)(e * 0.5f - lane * 0.02f); }
        __builtin_amdgcn_s_barrier();
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if (mode == 5) {
                a = *(const bf16x8*)(smem + ((it * 4 + wave) & 31) * 1024 + lane * 16);
                b = *(const bf16x8*)(smem + 32768 + ((it * 4 + wave) & 31) * 1024 + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (lane == 0) { cyc[blockIdx.x * 4 + wave] = t1 - t0; sink[blockIdx.x * 4 + wave] = s; }
    } else {
        __builtin_amdgcn_s_barrier();
        if (mode == 0) return;
        if (mode == 3 && wave != 4) return;
        const int n = iters * 2;                      // ~2 memory instructions per 16 MFMAs of the partner
        if (mode == 1 || mode == 3 || mode == 5) {
            for (int it = 0; it < n; ++it) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + 65536 + ((it * 4 + (wave - 4)) & 31) * 1024), 16,
                                                         (unsigned)(((it * 4 + wave) & 63) * 1024 + lane * 16), 0, 0, 0);
                if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (mode == 2) {
            f32x4 x = f32x4{0, 0, 0, 0};
            for (int it = 0; it < n; ++it) {
                const f32x4 v = *(const f32x4*)(smem + ((it * 4 + wave) & 63) * 1024 + lane * 16);
                x += v;
            }
            if (lane == 0) sink[4096 + blockIdx.x * 4 + wave - 4] = x[0];
        } else if (mode == 4) {
            u32x4 x = u32x4{0, 0, 0, 0};
            for (int it = 0; it < n; ++it) x += __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((it * 4 + wave) & 63) * 1024 + lane * 16), 0, 0);
            if (lane == 0) sink[4096 + blockIdx.x * 4 + wave - 4] = (float)x[0];
        }
    }
}

int main() {
    char* src; float* sink; unsigned long long* cyc;
    hipMalloc(&src, 65536); hipMemset(src, 0, 65536);
    hipMalloc(&sink, 65536 * 4); hipMalloc(&cyc, 1024 * 8);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int iters = 4000;
    unsigned long long h[1024];
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 131072, 0, src, sink, cyc, iters, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 1024 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
        const double per = s / 1024 / iters / 16;   // s_memtime ticks (100 MHz?) or cycles per MFMA
        printf("mode %d: %.2f ticks per MFMA (16x16x32 bf16; 16 cycles = back-to-back)\n", mode, per);
    }
    return 0;
}
