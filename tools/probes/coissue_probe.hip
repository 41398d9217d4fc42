// Does a VALU instruction issue under an in-flight MFMA of the SAME wave (one wave per SIMD)?  Cycles per MFMA for streams of
// [1 x v_mfma_f32_16x16x32_bf16 | 32x32x16, n x VALU] with the MFMA operands in AGPRs / VGPRs.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// MODE: 0 = all operands AGPR, 1 = A/B VGPR, C/D AGPR, 2 = all VGPR, 3 = A/B AGPR, C/D VGPR
template <int MODE, int NV, int KIND, int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink) {
    float x0 = threadIdx.x, x1 = 1.5f, x2 = 2.5f, x3 = 0.5f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 50; ++it) {
#define VALU_FMA "v_fma_f32 %0, %0, %1, %2\n"
#define VALU_EXP "v_exp_f32 %0, %0\n"
#define VALU_N ((NV >= 1 ? (KIND == 0 ? VALU_FMA : VALU_EXP) : "") )
        // the asm strings are assembled by the preprocessor per instantiation below
        if constexpr (SHAPE == 0) {
            if constexpr (MODE == 0) {
                REP64(asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], a[8:11], a[12:15], a[0:3]\n" ::: "a0", "a1", "a2", "a3");
                      if constexpr (NV >= 1) { if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); else asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                      if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));)
            } else if constexpr (MODE == 1) {
                REP64(asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], v[40:43], v[44:47], a[0:3]\n" ::: "a0", "a1", "a2", "a3", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
                      if constexpr (NV >= 1) { if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); else asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                      if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));)
            } else if constexpr (MODE == 2) {
                REP64(asm volatile("v_mfma_f32_16x16x32_bf16 v[48:51], v[40:43], v[44:47], v[48:51]\n" ::: "v48", "v49", "v50", "v51", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
                      if constexpr (NV >= 1) { if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); else asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                      if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));)
            } else {
                REP64(asm volatile("v_mfma_f32_16x16x32_bf16 v[48:51], a[8:11], a[12:15], v[48:51]\n" ::: "v48", "v49", "v50", "v51");
                      if constexpr (NV >= 1) { if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); else asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                      if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));)
            }
        } else {
            if constexpr (MODE == 0) {
                REP64(asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]\n" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
                      if constexpr (NV >= 1) { if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); else asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                      if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));)
            } else {
                REP64(asm volatile("v_mfma_f32_32x32x16_bf16 v[48:63], v[40:43], v[44:47], v[48:63]\n" ::: "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
                      if constexpr (NV >= 1) { if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); else asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); }
                      if constexpr (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2));
                      if constexpr (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(x1), "v"(x2));)
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (x0 + x3 == 12345.f) sink[0] = x0;
}

template <int MODE, int NV, int KIND, int SHAPE>
void run(const char* name, unsigned long long* d) {
    float* sink; hipMalloc(&sink, 4);
    hipLaunchKernelGGL((probe<MODE, NV, KIND, SHAPE>), dim3(256), dim3(256), 0, 0, d, sink);
    hipLaunchKernelGGL((probe<MODE, NV, KIND, SHAPE>), dim3(256), dim3(256), 0, 0, d, sink);
    hipDeviceSynchronize();
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-46s n_valu=%d %s : %6.1f cycles per MFMA\n", name, NV, KIND ? "exp+fma" : "fma", (double)h / (50.0 * 64));
    hipFree(sink);
}

#define SWEEP(MODE, SHAPE, NAME) \
    run<MODE, 0, 0, SHAPE>(NAME, d); run<MODE, 1, 0, SHAPE>(NAME, d); run<MODE, 2, 0, SHAPE>(NAME, d); run<MODE, 3, 0, SHAPE>(NAME, d); \
    run<MODE, 4, 0, SHAPE>(NAME, d); run<MODE, 1, 1, SHAPE>(NAME, d); run<MODE, 3, 1, SHAPE>(NAME, d);

int main() {
    unsigned long long* d; hipMalloc(&d, 8);
    SWEEP(0, 0, "16x16x32  A,B,C,D in AGPRs")
    SWEEP(1, 0, "16x16x32  A,B in VGPRs; C,D in AGPRs")
    SWEEP(2, 0, "16x16x32  A,B,C,D in VGPRs")
    SWEEP(3, 0, "16x16x32  A,B in AGPRs; C,D in VGPRs")
    SWEEP(0, 1, "32x32x16  A,B,C,D in AGPRs")
    SWEEP(2, 1, "32x32x16  A,B,C,D in VGPRs")
    return 0;
}
