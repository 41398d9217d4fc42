// Operand-layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3), exact small-integer data.
// Found (mx_scale_probe.hip + this file): lane l = (r = l&15, q = l>>4) holds A[row r][k = 16 q + 0..15] in dwords 0-3 and
// A[row r][k = 64 + 16 q + 0..15] in dwords 4-7 (byte order = k order), B likewise with its column on l&15 -- i.e. the two 16-byte
// reads of the bf16 16x16x32 k-steps 0 and 1 of a 128-byte row; the scale byte of lane l applies to row r, k in [32 q, 32 q + 32);
// C/D as the other 16x16 forms.  (With unit scales any k permutation shared by A and B passes: the scales pin the true order.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* C) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    v8i a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = *(const int*)(A + r * 128 + (j >> 2) * 64 + q * 16 + (j & 3) * 4);
        b[j] = *(const int*)(B + r * 128 + (j >> 2) * 64 + q * 16 + (j & 3) * 4);
    }
    const int xa = sa[r * 4 + q], xb = sb[r * 4 + q];
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, xa, 0, xb);
    for (int i = 0; i < 4; ++i) C[(q * 4 + i) * 16 + r] = c[i];
}

static uint8_t e4m3(int v) {   // small integers only: |v| <= 8
    static const uint8_t tab[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return v < 0 ? (uint8_t)(tab[-v] | 0x80) : tab[v];
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128], hsa[64], hsb[64];
    int iA[16 * 128], iB[16 * 128];
    srand(7);
    for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 9 - 4; iB[i] = rand() % 7 - 3; hA[i] = e4m3(iA[i]); hB[i] = e4m3(iB[i]); }
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 64; ++i) { hsa[i] = pass ? 125 + rand() % 5 : 127; hsb[i] = pass ? 126 + rand() % 3 : 127; }
        uint8_t *dA, *dB, *dsa, *dsb; float* dC; float hC[256];
        hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dC, sizeof hC);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, dsa, dsb, dC);
        hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int k = 0; k < 128; ++k)
                    ref += (double)iA[m * 128 + k] * iB[n * 128 + k] * ldexp(1.0, hsa[m * 4 + k / 32] - 127) * ldexp(1.0, hsb[n * 4 + k / 32] - 127);
                if (fabs(ref - hC[m * 16 + n]) > 1e-6 * fabs(ref) + 1e-6) { if (bad < 4) printf("pass %d C[%d][%d] = %g, expected %g\n", pass, m, n, hC[m * 16 + n], ref); ++bad; }
            }
        printf("pass %d (%s scales): %d mismatches of 256\n", pass, pass ? "varied" : "unit", bad);
    }
    return 0;
}
