// Probe (compile only: hipcc --offload-arch=gfx950 -O3 -save-temps): hipcc emits s_waitcnt vmcnt(0) at the first use of a plain
// register load when LDS-DMA loads issued AFTER it are still in flight (ideal: vmcnt(6)) -> side inputs of the DMA GEMMs are loaded from inline asm.
#include <hip/hip_runtime.h>
__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
__global__ void k(const float* a, const float* side, float* out, int n) {
    const int lane = threadIdx.x;
    float s[4];
    for (int i = 0; i < 4; ++i) s[i] = side[lane + 64 * i];          // plain loads (older)
    for (int i = 0; i < 6; ++i) dma16(a + lane * 4 + 256 * i, lds + 1024 * i);   // DMAs (younger)
    asm volatile("s_nop 0" ::: "memory");
    float r = s[0] + s[1] + s[2] + s[3];                              // consume plain loads: expect vmcnt(6) ideally
    out[lane] = r;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    out[lane + 64] = ((float*)lds)[lane];
}
