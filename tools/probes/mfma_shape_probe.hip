// Does the 2-stage NT main loop (256x256x64 tile, 8 waves as 2 x 4, each wave 128 x 64, 8 LDS-DMA pieces per wave issued at the
// top of a K-tile, fragment ring, one barrier per K-tile) run faster on v_mfma_f32_32x32x16_bf16 than on
// v_mfma_f32_16x16x32_bf16?  Same LDS bytes and flops per K-tile; the 32x32 form issues half as many MFMA instructions
// (an MFMA occupies the SIMD's issue port for 8 cycles either way), leaving more issue slots for the DMA pieces and reads.
// Not a GEMM: fragments read whatever the DMA brought.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define LDS_AS __attribute__((address_space(3)))
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void probe(const char* src, size_t region, float* sink, unsigned long long* cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const char* base = region > 65536 ? src + (size_t)blockIdx.x * region : src;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)region, 0x00020000);
    const unsigned wrap = (unsigned)region - 65536;
    float out = 0.f;
    unsigned goff = 0;
    auto dma = [&](int stage) {
        char* d = smem + stage * 65536 + wave * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(d + q * 1024), 16, goff + (wave * 8 + q) * 1024 + lane * 16, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(d + 32768 + q * 1024), 16, goff + (wave * 8 + 4 + q) * 1024 + lane * 16, 0, 0, 0);
        goff += 65536; if (goff > wrap) goff = 0;
    };
    unsigned long long t0 = 0, t1 = 0;
    if (SHAPE == 16) {
        f32x4 acc[8][4];
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        const int frow = lane & 15, fq = lane >> 4, fs = (lane >> 1) & 7;
        unsigned offA[2], offB[2];
        for (int ks = 0; ks < 2; ++ks) {
            const unsigned cb = (unsigned)(((ks * 4 + fq) ^ fs) << 4);
            offA[ks] = (unsigned)((wm * 128 + frow) * 128) + cb;
            offB[ks] = 32768u + (unsigned)((wn * 64 + frow) * 128) + cb;
        }
        dma(0);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int st = 0; st < steps; ++st) {
            dma((st + 1) & 1);
            const char* s = smem + (st & 1) * 65536;
            bf16x8 b[2][4], aq[3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j) b[ks][j] = *(const bf16x8*)(s + offB[ks] + j * 2048);
            aq[0] = *(const bf16x8*)(s + offA[0]);
            aq[1] = *(const bf16x8*)(s + offA[0] + 2048);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ks = t >> 3, i = t & 7;
                if (t + 2 < 16) aq[(t + 2) % 3] = *(const bf16x8*)(s + offA[(t + 2) >> 3] + ((t + 2) & 7) * 2048);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][j], aq[t % 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) out += acc[i][j][0] + acc[i][j][3];
    } else {
        f32x16 acc[4][2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        // A frag: row (base + lane % 32), 16-B chunk (2 ks + lane / 32) of the 8 in a 128-B row; same XOR swizzle
        const int r32 = lane & 31, kb = lane >> 5;
        unsigned offA[4], offB[4];
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned cb = (unsigned)((((ks * 2 + kb) ^ ((r32 >> 1) & 7))) << 4);
            offA[ks] = (unsigned)((wm * 128 + r32) * 128) + cb;
            offB[ks] = 32768u + (unsigned)((wn * 64 + r32) * 128) + cb;
        }
        dma(0);
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        for (int st = 0; st < steps; ++st) {
            dma((st + 1) & 1);
            const char* s = smem + (st & 1) * 65536;
            bf16x8 b[4][2], aq[3];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) b[ks][j] = *(const bf16x8*)(s + offB[ks] + j * 4096);
            aq[0] = *(const bf16x8*)(s + offA[0]);
            aq[1] = *(const bf16x8*)(s + offA[0] + 4096);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {                      // t = ks * 4 + i
                const int ks = t >> 2, i = t & 3;
                if (t + 2 < 16) aq[(t + 2) % 3] = *(const bf16x8*)(s + offA[(t + 2) >> 2] + ((t + 2) & 3) * 4096);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][j], aq[t % 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) out += acc[i][j][0] + acc[i][j][9];
    }
    if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t1 - t0; sink[blockIdx.x * 8 + wave] = out; }
}

template <int SHAPE>
void run(const char* name, const char* src, size_t region, float* sink, unsigned long long* cyc) {
    hipFuncSetAttribute((const void*)probe<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int steps = 1200;
    unsigned long long h[2048];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(256), dim3(512), 131072, 0, src, region, sink, cyc, steps);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE>), dim3(256), dim3(512), 131072, 0, src, region, sink, cyc, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
    const double flops = 256.0 * steps * 256 * 256 * 64 * 2;
    printf("%-24s %-9s %7.0f memtime ticks / K-tile; %.3f ms -> %6.0f TFLOP/s equivalent\n", name, region > 65536 ? "streaming" : "L2-hot",
           s / 2048 / steps, ms, flops / ms / 1e9);
}

int main() {
    const size_t big = 8u << 20;
    char* src; float* sink; unsigned long long* cyc;
    hipMalloc(&src, big * 256);
    {   // random bf16 operands in [-1, 1)
        const size_t n = 32u << 20;
        unsigned short* h = (unsigned short*)malloc(n * 2);
        unsigned x = 12345u;
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; const float f = (float)(int)(x >> 8) / 8388608.0f - 1.0f; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
        for (size_t off = 0; off < big * 256; off += n * 2) hipMemcpy(src + off, h, n * 2, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&sink, 65536 * 4); hipMalloc(&cyc, 2048 * 8);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t region : {(size_t)131072, big}) {
            run<16>("16x16x32 (64 MFMA/tile)", src, region, sink, cyc);
            run<32>("32x32x16 (32 MFMA/tile)", src, region, sink, cyc);
        }
    return 0;
}
