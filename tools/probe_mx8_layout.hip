#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// raw: lane l supplies a[8], b[8] dwords and scale dwords from arrays indexed by lane
__global__ void k_raw(const int* A, const int* B, const int* sA, const int* sB, float* D, int opa, int opb) {
    const int l = threadIdx.x;
    v8i a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[l * 8 + j]; b[j] = B[l * 8 + j]; }
    v16f c; for (int j = 0; j < 16; ++j) c[j] = 0.f;
    if (opa == 0 && opb == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sA[l], 0, sB[l]);
    else if (opa == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sA[l], 0, sB[l]);
    else if (opa == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, sA[l], 0, sB[l]);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, sA[l], 0, sB[l]);
    for (int j = 0; j < 16; ++j) D[l * 16 + j] = c[j];
}
int main() {
    int hA[512], hB[512], hsA[64], hsB[64]; float hD[1024];
    int *dA, *dB, *dsA, *dsB; float* dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsA, 256); hipMalloc(&dsB, 256); hipMalloc(&dD, 4096);
    auto run = [&](int opa) {
        hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dsA, hsA, 256, hipMemcpyHostToDevice); hipMemcpy(dsB, hsB, 256, hipMemcpyHostToDevice);
        k_raw<<<1, 64>>>(dA, dB, dsA, dsB, dD, opa, 0); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    };
    for (int i = 0; i < 64; ++i) { hsA[i] = 0x7f7f7f7f; hsB[i] = 0x7f7f7f7f; }
    // (1) which output (lane, reg) does A-lane LA byte JA x B-lane LB byte JB feed?  one-hot 1.0 (0x38) in A and all-ones B
    printf("== A one-hot (lane, byte) vs all-ones B: nonzero outputs -> A row; unit scales\n");
    for (int LA : {0, 1, 31, 32, 33, 63}) for (int JA : {0, 1, 15, 16, 31}) {
        memset(hA, 0, sizeof(hA)); for (int i = 0; i < 512; ++i) hB[i] = 0x38383838;
        ((uint8_t*)hA)[LA * 32 + JA] = 0x38; run(0);
        int cnt = 0, l0 = -1, j0 = -1; float v0 = 0; for (int i = 0; i < 1024; ++i) if (hD[i] != 0) { if (!cnt) { l0 = i / 16; j0 = i % 16; v0 = hD[i]; } ++cnt; }
        printf("  A(lane %2d, byte %2d): %d nonzero outputs, first at (lane %d, reg %d) = %g  -> row %d\n", LA, JA, cnt, l0, j0, v0, (j0 & 3) + 8 * (j0 >> 2) + 4 * (l0 >> 5));
    }
    // (2) k pairing: A one-hot at (LA, JA); B one-hot at (LB, JB): nonzero iff same k
    printf("== k pairing: for A(lane 0, byte JA) find the B (lane in {0, 32}, byte) that pairs\n");
    for (int LA : {0, 32}) for (int JA : {0, 1, 4, 15, 16, 17, 31}) {
        memset(hA, 0, sizeof(hA)); ((uint8_t*)hA)[LA * 32 + JA] = 0x38;
        for (int LB : {0, 32}) for (int JB = 0; JB < 32; ++JB) {
            memset(hB, 0, sizeof(hB)); ((uint8_t*)hB)[LB * 32 + JB] = 0x38; run(0);
            float s = 0; for (int i = 0; i < 1024; ++i) s += fabsf(hD[i]);
            if (s != 0) printf("  A(lane %2d, byte %2d) pairs with B(lane %2d, byte %2d)\n", LA, JA, LB, JB);
        }
    }
    // (3) scales: A data 1.0 ONLY in lanes of half HA (other half 0), B all ones -> every row sums 32; one A scale byte = 128
    printf("== scales: A = 1.0 in one lane half only (row sum 32); one A scale byte set to 128 (x2 -> 64 if it applies to that half)\n");
    for (int i = 0; i < 512; ++i) hB[i] = 0x38383838;
    for (int HA : {0, 1}) for (int opa : {0, 1, 2, 3}) for (int L : {5, 37}) for (int bt : {0, 1, 2, 3}) {
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) hA[l * 8 + j] = (l >> 5) == HA ? 0x38383838 : 0;
        for (int i = 0; i < 64; ++i) hsA[i] = 0x7f7f7f7f;
        hsA[L] = (hsA[L] & ~(0xff << (8 * bt))) | (128 << (8 * bt)); run(opa);
        int cnt = 0; float val = 0; int row0 = -1;
        for (int i = 0; i < 1024; ++i) if (hD[i] != 32.f) { int l = i / 16, j = i % 16; if (!cnt) { row0 = (j & 3) + 8 * (j >> 2) + 4 * (l >> 5); val = hD[i]; } ++cnt; }
        if (cnt) printf("  data half %d, opsel_a %d, scale lane %2d byte %d = 128: %d outputs changed, row %d -> %g\n", HA, opa, L, bt, cnt, row0, val);
    }
    return 0;
}
