// v_permlane32_swap semantics on gfx950: which halves move where?   hipcc --offload-arch=gfx950 -O2 -o permlane_probe permlane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 512); k<<<1, 64>>>(d);
    unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("r0: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
    printf("r1: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
