// v_dot2_f32_bf16 + DPP row reductions over 8 lanes: semantics check.   hipcc --offload-arch=gfx950 -O2 -o dot2_dpp_probe dot2_dpp_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    bf16x2 a, b;
    a[0] = (__bf16)(float)(lane + 1); a[1] = (__bf16)2.0f;
    b[0] = (__bf16)3.0f; b[1] = (__bf16)(float)(lane & 3);
    float d = __builtin_amdgcn_fdot2_f32_bf16(a, b, 100.0f, false);      // (lane + 1) * 3 + 2 * (lane & 3) + 100
    out[lane] = d;
    float s = (float)lane;
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x4E, 0xF, 0xF, true));
    s += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x141, 0xF, 0xF, true));
    out[64 + lane] = s;                                                  // sum of the lane's group of 8: 28 + 64 * (lane / 8)
}
int main() {
    float* d; hipMalloc(&d, 512); k<<<1, 64>>>(d);
    float h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i) printf("lane %d: dot2 %.1f (expect %.1f)  sum8 %.1f (expect %.1f)\n", i, h[i], (i + 1) * 3.0f + 2.0f * (i & 3) + 100, h[64 + i], 28.0f + 64 * (i / 8));
    return 0;
}
