// v_mfma_f64_16x16x4_f64 issue rate: cycles per instruction for one wave, and chip-level TFLOP/s
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, int iters, double a0) {
    v4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4{0., 0., 0., 0.};
    double a = a0 + threadIdx.x * 1e-3, b = a0 - threadIdx.x * 2e-3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    struct { int blocks, threads; const char* name; } cfgs[] = {{1, 64, "1 wave"}, {1, 256, "4 waves, 1 CU"}, {256, 256, "1 WG/CU"}, {1024, 256, "4 WG/CU"}, {2048, 256, "8 WG/CU"}};
    for (auto& c : cfgs) {
        k<<<c.blocks, c.threads>>>(out, cyc, 100, 0.5);
        hipEventRecord(e0); k<<<c.blocks, c.threads>>>(out, cyc, iters, 0.5); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        double waves = double(c.blocks) * c.threads / 64;
        double tf = waves * iters * 8.0 * 2048.0 / (ms * 1e-3) / 1e12;
        printf("%-16s cycles/MFMA (wave view) %.1f   %.2f TFLOP/s   wall %.3f ms  implied clock %.2f GHz\n", c.name, double(h) / (iters * 8.0), tf, ms, double(h) / (ms * 1e-3) / 1e9);
    }
    return 0;
}
