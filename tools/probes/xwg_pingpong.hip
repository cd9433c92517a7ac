// Cross-workgroup hand-off latency on gfx950: two workgroups exchange a sequence number through global memory with relaxed
// agent-scope atomic loads / stores (sc1), the primitive the multi-CU coordinate-descent team uses.  Prints cycles and ns per
// round trip for a partner on the same XCD (workgroups 0 and 8 of the launch) and on another XCD (0 and 1).
// Build: hipcc --offload-arch=gfx950 -O3 xwg_pingpong.hip -o xwg_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned long long ld(unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_pingpong(unsigned long long *box, int partner, int iters, unsigned long long *out, int sleep) {
    const int wg = blockIdx.x;
    if (wg != 0 && wg != partner) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[4 + (wg != 0)] = xcc & 0xf;
    unsigned long long *ping = box, *pong = box + 64;  // separate cache lines
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    long spins = 0;
    for (int i = 1; i <= iters; ++i) {
        if (wg == 0) {
            if (threadIdx.x == 0) st(ping, i);
            while (ld(pong) < (unsigned long long)i) {
                if (sleep) __builtin_amdgcn_s_sleep(1);
                if (++spins > (1l << 26)) return;
            }
        } else {
            while (ld(ping) < (unsigned long long)i) {
                if (sleep) __builtin_amdgcn_s_sleep(1);
                if (++spins > (1l << 26)) return;
            }
            if (threadIdx.x == 0) st(pong, i);
        }
    }
    if (wg == 0 && threadIdx.x == 0) {
        out[0] = __builtin_readcyclecounter() - t0;
        out[1] = wall_clock64() - w0;  // 100 MHz
        out[2] = spins;
    }
}

int main() {
    unsigned long long *box, *out, h[8];
    hipMalloc(&box, 4096);
    hipMalloc(&out, 64);
    const int iters = 20000;
    for (int sleep = 0; sleep < 2; ++sleep)
        for (int partner : {8, 1, 4, 16}) {
            hipMemset(box, 0, 4096);
            hipMemset(out, 0, 64);
            k_pingpong<<<32, 64>>>(box, partner, iters, out, sleep);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
            hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            printf("partner wg %2d (xcc %llu vs %llu) sleep=%d: %.0f cycles, %.0f ns per round trip (%.1f polls)\n", partner, h[4], h[5],
                   sleep, double(h[0]) / iters, double(h[1]) * 10.0 / iters, double(h[2]) / iters);
        }
    return 0;
}
