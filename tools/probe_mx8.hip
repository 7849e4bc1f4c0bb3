// Probe of the gfx950 MX-scaled fp8 MFMA and the fp8 convert (run on the GPU box): checks the operand / scale layout assumed by
// csrc/gemm_mx8.h against a host reference.  hipcc --offload-arch=gfx950 -O2 probe_mx8.hip -o probe_mx8 && ./probe_mx8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k_mfma(const uint8_t* A, const uint8_t* B, const uint8_t* sA, const uint8_t* sB, float* D) {
    // A: [32 rows][64 k] fp8 e4m3, B: [32 cols][64 k]; sA/sB: [32][2] E8M0 scale bytes (one per 32-k block)
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i a, b;
    for (int j = 0; j < 8; ++j) { a[j] = ((const int*)(A + r * 64 + h * 32))[j]; b[j] = ((const int*)(B + r * 64 + h * 32))[j]; }
    // scale dword: bytes {blk0, blk1, x, x} of this lane's row, pre-shifted so that byte 0 is the lane's own block
    int sa = (int)((uint32_t)(sA[r * 2] | (sA[r * 2 + 1] << 8)) >> (8 * h)), sb = (int)((uint32_t)(sB[r * 2] | (sB[r * 2 + 1] << 8)) >> (8 * h));
    v16f c; for (int j = 0; j < 16; ++j) c[j] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int j = 0; j < 16; ++j) D[((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = c[j];
}
__global__ void k_cvt(const float* x, uint8_t* y, int n) {
    int i = blockIdx.x * 64 + threadIdx.x; if (2 * i + 1 >= n + 1) return;
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    y[2 * i] = pk & 0xff; y[2 * i + 1] = (pk >> 8) & 0xff;
}
static float e4m3(uint8_t v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; if (e == 15 && m == 7) return NAN; float f = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6); return s ? -f : f; }
static uint8_t enc(float f) {      // RNE, saturate to 448
    uint8_t s = f < 0 ? 0x80 : 0; float a = fabsf(f); if (!(a == a)) return 0x7f; if (a > 448.f) a = 448.f;
    uint8_t best = 0; float bd = 1e30f;
    for (int v = 0; v < 0x7f; ++v) { float d = fabsf(e4m3((uint8_t)v) - a); if (d < bd || (d == bd && !(v & 1))) { bd = d; best = (uint8_t)v; } }
    return s | best;
}
int main() {
    uint8_t hA[32 * 64], hB[32 * 64], hsA[64], hsB[64]; float hD[1024];
    srand(1);
    for (int i = 0; i < 2048; ++i) { hA[i] = rand() & 0xff; if ((hA[i] & 0x7f) == 0x7f) hA[i] = 0x3c; hB[i] = rand() & 0xff; if ((hB[i] & 0x7f) == 0x7f) hB[i] = 0x41; }
    for (int i = 0; i < 64; ++i) { hsA[i] = 120 + rand() % 12; hsB[i] = 122 + rand() % 9; }
    uint8_t *dA, *dB, *dsA, *dsB; float* dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice); hipMemcpy(dsA, hsA, 64, hipMemcpyHostToDevice); hipMemcpy(dsB, hsB, 64, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dsA, dsB, dD); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0, ref_max = 0;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        double s = 0;
        for (int k = 0; k < 64; ++k) s += (double)e4m3(hA[m * 64 + k]) * ldexp(1.0, hsA[m * 2 + k / 32] - 127) * (double)e4m3(hB[n * 64 + k]) * ldexp(1.0, hsB[n * 2 + k / 32] - 127);
        double d = fabs(s - hD[m * 32 + n]); if (d > worst) worst = d; if (fabs(s) > ref_max) ref_max = fabs(s);
    }
    printf("mfma_scale_f32_32x32x64_f8f6f4: max |err| %.3e (|ref| up to %.3e) -> layout %s\n", worst, ref_max, worst < 1e-3 * ref_max ? "CONFIRMED (row = lane&31, k = 32*(lane>>5)+j, own-lane scale byte 0)" : "MISMATCH");
    const int n = 4096; float hx[n]; uint8_t hy[n];
    for (int i = 0; i < n; ++i) hx[i] = (i < 16 ? (float[]){0.f, 1.f, -1.f, 448.f, 449.f, 1000.f, -1e6f, 0.001f, 0.0009765625f, 1.0625f, 1.1875f, 17.f, 18.f, 19.f, 2.5e-4f, -460.f}[i] : ldexpf((rand() / (float)RAND_MAX) * 2 - 1, rand() % 20 - 10));
    float* dx; uint8_t* dy; hipMalloc(&dx, n * 4); hipMalloc(&dy, n); hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k_cvt<<<n / 128, 64>>>(dx, dy, n); hipMemcpy(hy, dy, n, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) if (hy[i] != enc(hx[i]) && !(hx[i] == 0.f && (hy[i] & 0x7f) == 0)) { if (bad < 8) printf("  cvt mismatch x=%g hw=0x%02x (%g) sw=0x%02x (%g)\n", hx[i], hy[i], e4m3(hy[i]), enc(hx[i]), e4m3(enc(hx[i]))); ++bad; }
    printf("cvt_pk_fp8_f32 (OCP e4m3, RNE, saturating): %d / %d mismatches vs software encoder\n", bad, n);
    return 0;
}
