// Which XCD does block b of a launch run on?  s_getreg_b32 hwreg(HW_REG_XCC_ID) against b % 8.
// Build: hipcc --offload-arch=gfx950 -O3 xcc_probe.hip -o xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    const int n = 64;
    int *d, h[n];
    hipMalloc(&d, n * 4);
    k<<<n, 64>>>(d);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%d:%#x ", i, h[i]);
    printf("\n");
    return 0;
}
