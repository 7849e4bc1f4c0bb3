// Probe: semantics of ds_read_b64_tr_b16 (gfx950).  LDS holds lds[i] = i (u16).  Each lane supplies a byte address;
// we print, per lane, the four 16-bit values it receives.  Build: hipcc --offload-arch=gfx950 probe_tr16.hip -o probe_tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int mode, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int l = threadIdx.x;
    int elem;                                   // element index (u16 units) this lane points at
    if (mode == 0) elem = (l & 15) * 4 + (l >> 4) * 64;           // 16 lanes cover 64 contiguous elements
    else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;      // each lane its own row (stride 64), 4 contiguous
    else elem = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 16; // [4 rows][16 cols] block per 16-lane group, row stride 64
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + elem));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    return 0;
}
