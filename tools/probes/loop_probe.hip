// GEMM main-loop structure probe (MI355X): cycles per 32-deep K-step of a 256x256 tile for different ways of
// feeding the LDS ring.  No real GEMM: fragment reads hit arbitrary ring data, DMA source is a 64 KiB L2-hot region
// (mode bit 8: a large streaming region instead).  8 MFMA waves (128 accumulator VGPRs each, 12 ds_read_b128 + 32 MFMA
// per step) and, optionally, 4 loader waves.
//   variant 0: no DMA at all                         (upper bound)
//   variant 1: each MFMA wave issues its 4 DMA pieces at the top of the step, vmcnt(12) + barrier   (ring kernel)
//   variant 2: 4 loader waves issue 8 pieces each per step (vmcnt(24) + barrier), MFMA waves: reads + MFMA + barrier
//   variant 3: as 1 but DMA pieces spread between MFMA groups
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define LDS_AS __attribute__((address_space(3)))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VARIANT, int NTHREADS>
__global__ __launch_bounds__(NTHREADS, NTHREADS == 768 ? 3 : 2) void probe(const char* src, size_t region, float* sink,
                                                                          unsigned long long* cyc, int steps, int stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = region > 65536 ? src + (size_t)blockIdx.x * region : src;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)region, 0x00020000);
    const unsigned wrap = (unsigned)region - 32768;
    if (wave < 8) {
        const int wm = wave >> 2, wn = wave & 3;
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        const unsigned offA = (wm * 128 + (lane & 15)) * 64 + ((lane >> 4) ^ ((4 - (((lane & 15) >> 2) & 3)) & 3)) * 16;
        const unsigned offB = 16384 + (wn * 64 + (lane & 15)) * 64 + ((lane >> 4) ^ ((4 - (((lane & 15) >> 2) & 3)) & 3)) * 16;
        __builtin_amdgcn_s_barrier();
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        unsigned goff = 0;
        for (int st = 0; st < steps; ++st) {
            const char* slot = smem + (st % 5) * 32768;
            char* dst = smem + ((st + 4) % 5) * 32768 + wave * 2048;
            if (VARIANT == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + (q >> 1) * 16384 + (q & 1) * 1024), 16,
                                                             goff + (wave * 4 + q) * 1024 + lane * 16, 0, 0, 0);
            }
            bf16x8 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(slot + offB + j * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bf16x8 a = *(const bf16x8*)(slot + offA + i * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a, acc[i][j], 0, 0, 0);
                if (VARIANT == 3 && (i & 1))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + (i >> 2) * 16384 + ((i >> 1) & 1) * 1024), 16,
                                                             goff + (wave * 4 + (i >> 1)) * 1024 + lane * 16, 0, 0, 0);
            }
            goff += 32768; if (goff > wrap) goff = 0;
            if (VARIANT == 1 || VARIANT == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float s = 0;
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][2];
        if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; sink[blockIdx.x * 8 + wave] = s; }
    } else {                                          // loader waves (VARIANT 2)
        const int lw = wave - 8;
        __builtin_amdgcn_s_barrier();
        unsigned goff = 0;
        for (int st = 0; st < steps; ++st) {
            char* dst = smem + ((st + 4) % 5) * 32768 + lw * 8192;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(dst + q * 1024), 16,
                    stride ? (unsigned)((((lw * 8 + q) * 16 + (lane >> 2)) * stride + (lane & 3) * 16 + (goff >> 9)) % wrap) : goff + (lw * 8 + q) * 1024 + lane * 16, 0, 0, 0);
            goff += 32768; if (goff > wrap) goff = 0;
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <int V, int NT>
void run(const char* name, const char* src, size_t region, float* sink, unsigned long long* cyc) {
    const int stride = getenv("PROBE_STRIDE") ? atoi(getenv("PROBE_STRIDE")) : 0;
    hipFuncSetAttribute((const void*)probe<V, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    const int steps = 2400;
    unsigned long long h[2048];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<V, NT>), dim3(256), dim3(NT), 163840, 0, src, region, sink, cyc, steps, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<V, NT>), dim3(256), dim3(NT), 163840, 0, src, region, sink, cyc, steps, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
    const double per_step = s / 2048 / steps;
    const double flops = 256.0 * steps * 256 * 256 * 32 * 2;
    printf("%-58s %-9s %7.0f cycles/step -> MFMA pipe %5.1f %% busy; %.3f ms -> %.2f GHz, %6.0f TFLOP/s equivalent\n", name,
           region > 65536 ? "streaming" : "L2-hot", per_step, 100.0 * 1024 / per_step, ms, per_step * steps / ms / 1e6, flops / ms / 1e9);
}

int main() {
    const size_t big = 8u << 20;
    char* src; float* sink; unsigned long long* cyc;
    hipMalloc(&src, big * 256); hipMemset(src, 0, big * 256);
    if (getenv("PROBE_RANDOM")) {       // random bf16 operands in [-1, 1): zero data lets the chip hold a higher clock
        const size_t n = 32u << 20;
        unsigned short* h = (unsigned short*)malloc(n * 2);
        unsigned x = 12345u;
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; const float f = (float)(int)(x >> 8) / 8388608.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
        for (size_t off = 0; off < big * 256; off += n * 2) hipMemcpy(src + off, h, n * 2, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&sink, 65536 * 4); hipMalloc(&cyc, 2048 * 8);
    for (size_t region : {(size_t)65536, big}) {
        run<0, 512>("0 no DMA", src, region, sink, cyc);
        run<1, 512>("1 every MFMA wave issues 4 pieces at the top of the step", src, region, sink, cyc);
        run<3, 512>("3 every MFMA wave, pieces spread between MFMA groups", src, region, sink, cyc);
        run<2, 768>("2 four loader waves issue 8 pieces each", src, region, sink, cyc);
    }
    return 0;
}
