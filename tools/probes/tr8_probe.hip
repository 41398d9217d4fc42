// What does ds_read_b64_tr_b8 deliver?  LDS image: 32 rows x 16 bytes, byte (r, c) = r * 16 + c (r < 16) or 0x80 | ... for the second
// half.  Pattern A: lane i of a 16-lane group supplies the address of row (i >> 1), columns 8 (i & 1) .. + 7  (the 8-bit analogue of
// the tr_b16 rule: 16 lanes x 8 B = an 8-row x 16-column block).  Pattern B: lane i supplies row i, columns 0..7 (a 16 x 8 block).
// Prints, per lane, the 8 bytes received.  Build: hipcc --offload-arch=gfx950 -O2 tr8_probe.hip -o tr8_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void probe(uint32_t* out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint8_t img[64 * 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 16; i += 64) img[i] = (uint8_t)i;          // rows 0..15 distinct; rows 16.. repeat mod 256
    __syncthreads();
    const int g = lane >> 4, i = lane & 15;
    uint32_t addr;
    if (pattern == 0) addr = (uint32_t)((g * 8 + (i >> 1)) * 16 + (i & 1) * 8);     // group g: rows 8g..8g+7 (g >= 2 wraps values)
    else addr = (uint32_t)(((g & 1) * 16 + i) * 16 + (g >> 1) * 8);
    addr += (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)img;
    int __attribute__((ext_vector_type(2))) v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    out[lane * 2] = v[0];
    out[lane * 2 + 1] = v[1];
}

int main() {
    uint32_t* d;
    hipMalloc(&d, 64 * 8);
    for (int pat = 0; pat < 2; ++pat) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pat);
        uint32_t h[128];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) {
                const unsigned v = (h[l * 2 + (b >> 2)] >> (8 * (b & 3))) & 255u;
                printf(" (%2u,%2u)", v >> 4, v & 15);
            }
            printf("\n");
        }
    }
    return 0;
}
