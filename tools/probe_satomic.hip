// probe (round-2 groundwork): s_atomic_add — a scalar, lgkmcnt-tracked atomic — works on gfx950 (4096 distinct tickets on MI355X), so a
// persistent kernel can claim tiles dynamically without touching the vmcnt accounting of its DMA ring.  hipcc --offload-arch=gfx950 -O3 probe_satomic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(int* ctr, int* out) {
    int r;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(ctr), "0"(1) : "memory");
    if (threadIdx.x == 0) out[blockIdx.x] = r;
}
int main() {
    int *ctr, *out; const int N = 4096;
    hipMalloc(&ctr, 4); hipMalloc(&out, N * 4); hipMemset(ctr, 0, 4);
    hipLaunchKernelGGL(k, dim3(N), dim3(64), 0, 0, ctr, out);
    std::vector<int> h(N); int c = -1;
    hipMemcpy(h.data(), out, N * 4, hipMemcpyDeviceToHost); hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = c == N; for (int i = 0; i < N; ++i) ok = ok && h[i] == i;
    printf("counter %d, tickets distinct 0..N-1: %s\n", c, ok ? "yes" : "NO");
    return 0;
}
