// Issue cost of filler instructions beside MFMAs, one wave per SIMD.  Stream: [MFMA, n x filler]; cycles per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

// KIND: 0 fma, 1 exp, 2 pk_mul, 3 cvt_pk_bf16, 4 ds_read_b128, 5 ds_read_b64_tr_b16, 6 ds_write_b64, 7 v_max3, 8 v_pk_add, 9 v_accvgpr_read
template <int KIND>
__device__ __forceinline__ void filler(float& x0, float& x1, f2& p0, f2& p1, i4& d4, i2& d2, unsigned a) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1));
    if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x0));
    if constexpr (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));
    if constexpr (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(x0) : "v"(x1));
    if constexpr (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(d4) : "v"(a));
    if constexpr (KIND == 5) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d2) : "v"(a));
    if constexpr (KIND == 6) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(d2));
    if constexpr (KIND == 7) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(x1));
    if constexpr (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(p1));
    if constexpr (KIND == 9) asm volatile("v_accvgpr_read_b32 %0, a20" : "=v"(x0));
}

template <int KIND, int NV, int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, float* sink) {
    __shared__ char lds[65536];
    float x0 = threadIdx.x, x1 = 1.5f;
    f2 p0 = {1.f, 2.f}, p1 = {1.0001f, 0.9999f};
    i4 d4 = {0, 0, 0, 0}; i2 d2 = {0, 0};
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 50; ++it) {
        REP64(if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 a[0:3], a[8:11], a[12:15], a[0:3]" ::: "a0", "a1", "a2", "a3");
              else asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], a[16:19], a[20:23], a[0:15]" ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
              if constexpr (NV >= 1) filler<KIND>(x0, x1, p0, p1, d4, d2, a);
              if constexpr (NV >= 2) filler<KIND>(x0, x1, p0, p1, d4, d2, a);
              if constexpr (NV >= 3) filler<KIND>(x0, x1, p0, p1, d4, d2, a);
              if constexpr (NV >= 4) filler<KIND>(x0, x1, p0, p1, d4, d2, a);
              if constexpr (NV >= 6) { filler<KIND>(x0, x1, p0, p1, d4, d2, a); filler<KIND>(x0, x1, p0, p1, d4, d2, a); })
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (x0 + p0[0] + d4[0] + d2[0] == 12345.f) sink[0] = x0;
}

template <int KIND, int NV, int SHAPE>
double run(unsigned long long* d) {
    float* sink; hipMalloc(&sink, 4);
    hipLaunchKernelGGL((probe<KIND, NV, SHAPE>), dim3(256), dim3(256), 0, 0, d, sink);
    hipLaunchKernelGGL((probe<KIND, NV, SHAPE>), dim3(256), dim3(256), 0, 0, d, sink);
    hipDeviceSynchronize();
    unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipFree(sink);
    return (double)h / (50.0 * 64);
}
template <int KIND>
void sweep(const char* name, unsigned long long* d) {
    printf("%-22s 16x16x32: n=0 %5.1f  1 %5.1f  2 %5.1f  3 %5.1f  4 %5.1f  6 %5.1f | 32x32x16: n=0 %5.1f  1 %5.1f  2 %5.1f  3 %5.1f  4 %5.1f  6 %5.1f\n", name,
           run<KIND, 0, 0>(d), run<KIND, 1, 0>(d), run<KIND, 2, 0>(d), run<KIND, 3, 0>(d), run<KIND, 4, 0>(d), run<KIND, 6, 0>(d),
           run<KIND, 0, 1>(d), run<KIND, 1, 1>(d), run<KIND, 2, 1>(d), run<KIND, 3, 1>(d), run<KIND, 4, 1>(d), run<KIND, 6, 1>(d));
}
int main() {
    unsigned long long* d; hipMalloc(&d, 8);
    sweep<0>("v_fma_f32", d); sweep<1>("v_exp_f32", d); sweep<2>("v_pk_mul_f32", d); sweep<8>("v_pk_add_f32", d); sweep<3>("v_cvt_pk_bf16_f32", d);
    sweep<7>("v_max3_f32", d); sweep<9>("v_accvgpr_read_b32", d); sweep<4>("ds_read_b128", d); sweep<5>("ds_read_b64_tr_b16", d); sweep<6>("ds_write_b64", d);
    return 0;
}
