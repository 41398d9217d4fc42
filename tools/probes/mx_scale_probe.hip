// Which (row, k-block) does the scale byte of lane L apply to?  A = ones; B = ones in k-block kb only; lane L carries scale 2.0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const uint8_t* A, const uint8_t* B, const int* sa, const int* sb, float* C) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    v8i a, b;
    for (int j = 0; j < 8; ++j) { a[j] = *(const int*)(A + r * 128 + q * 32 + j * 4); b[j] = *(const int*)(B + r * 128 + q * 32 + j * 4); }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 4; ++i) C[(q * 4 + i) * 16 + r] = c[i];
}
int main() {
    uint8_t hA[2048], hB[2048]; int hsa[64], hsb[64]; float hC[256];
    uint8_t *dA, *dB; int *dsa, *dsb; float* dC;
    (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dC, 1024);
    for (int which = 0; which < 2; ++which) {          // 0: scale A varies, 1: scale B varies
        printf("%s scale: lane -> (row/col, k-block) [value]\n", which ? "B" : "A");
        for (int L = 0; L < 64; ++L) {
            int found = 0;
            for (int kb = 0; kb < 4; ++kb) {
                for (int i = 0; i < 2048; ++i) { hA[i] = 0x38; hB[i] = 0x38; }
                uint8_t* masked = which ? hA : hB;      // the OTHER operand selects the k-block
                for (int rr = 0; rr < 16; ++rr) for (int k = 0; k < 128; ++k) if (k / 32 != kb) masked[rr * 128 + k] = 0;
                for (int i = 0; i < 64; ++i) { hsa[i] = 127; hsb[i] = 127; }
                (which ? hsb : hsa)[L] = 128;
                (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
                (void)hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
                probe<<<1, 64>>>(dA, dB, dsa, dsb, dC);
                (void)hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
                for (int m = 0; m < 16; ++m) {
                    const float v = which ? hC[0 * 16 + m] : hC[m * 16 + 0];      // column m (B) or row m (A)
                    if (v != 32.f) { printf("  lane %2d -> (%2d, %d) [%g]", L, m, kb, v); ++found; }
                }
            }
            printf(found ? "\n" : "  lane %2d -> nothing\n", L);
        }
    }
    return 0;
}
