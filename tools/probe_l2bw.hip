// Probe: what does one CU / the whole chip sustain L2 -> CU?  (the persistent GEMMs measure ~20 B/clk/CU of LDS-DMA fill; is that the
// kernel or the cache?)  Every workgroup streams a region that is shared by the workgroups of its XCD (blockIdx & 7), sized to sit in
// that XCD's L2 (1 MiB), in the Infinity Cache (16 MiB per XCD = 128 MiB) or in HBM (256 MiB per XCD), with plain 16-byte loads
// (mode 0) or LDS-DMA (mode 1), at 4 / 8 / 16 waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_l2bw.hip -o tools/_bin/probe_l2bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

// region bytes per XCD = rb (power of two); every wave walks the region in 1 KiB pieces, piece index advancing by the number of waves
// of the XCD, for `iters` rounds of 8 pieces
template <int MODE>
__global__ void __launch_bounds__(1024) stream_kernel(const unsigned char* base, size_t rb, int iters, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    const unsigned char* reg = base + (size_t)xcd * rb;
    const size_t pieces = rb >> 10, stride = (size_t)nwg * nw;
    size_t pc = (size_t)wg * nw + wave;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = *reinterpret_cast<const uint4*>(reg + ((pc & (pieces - 1)) << 10) + lane * 16); pc += stride; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += __uint_as_float(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w);
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) { dma16(reg + ((pc & (pieces - 1)) << 10) + lane * 16, lds + (size_t)wave * 16384 + ((it & 1) * 8 + u) * 1024); pc += stride; }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the previous round has landed; this one stays in flight
        }
    }
    if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc = ((float*)lds)[threadIdx.x]; }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t total = (size_t)8 * (256u << 20);
    unsigned char* buf; float* out;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 1, total));
    CK(hipFuncSetAttribute((const void*)stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t regions[] = {256u << 10, 1u << 20, 2u << 20, 16u << 20, 256u << 20};
    for (int mode = 0; mode < 2; ++mode)
        for (size_t rb : regions)
            for (int waves : {4, 8, 16}) {
                if (mode == 1 && waves > 8) continue;             // 16 KiB of LDS per wave
                const int iters = 2048;
                const double bytes = 256.0 * waves * iters * 8 * 1024;
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) stream_kernel<0><<<256, waves * 64, 0>>>(buf, rb, iters, out);
                    else stream_kernel<1><<<256, waves * 64, waves * 16384>>>(buf, rb, iters, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                }
                const double tbs = bytes / (best * 1e-3) / 1e12;
                printf("%-9s region/XCD %6zu KiB  %2d waves/CU : %7.2f TB/s  = %5.1f B/clk/CU @2.4GHz\n", mode ? "lds-dma" : "vgpr", rb >> 10, waves, tbs,
                       tbs * 1e12 / 256 / 2.4e9);
            }
    return 0;
}
