// single-wave latency micro-benchmarks (gfx950): cycles per dependent instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double a, double b, int iters) {
    double x = a + threadIdx.x * 1e-9, y = b;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) x = __builtin_fma(x, y, b);                       // dependent fma f64
            if (MODE == 1) x = x + y;                                         // dependent add f64
            if (MODE == 2) x = __builtin_fmax(x + y, 0.5);                    // add + max
            if (MODE == 3) {                                                  // readlane round trip + fma
                int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
                x = __builtin_fma(__hiloint2double(hi, lo), y, b);
            }
            if (MODE == 4) x = x * y;                                         // dependent mul
            if (MODE == 5) { float f = (float)x; f = __builtin_fmaf(f, 1.0001f, 0.5f); x = f; }  // cvt chain
            if (MODE == 6) x = b / (x + y);                                   // division chain
            if (MODE == 7) x = __builtin_copysign(__builtin_fmax(__builtin_fabs(x) - y, 0.0), x) * b + 2.0; // threshold chain
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// independent streams: issue rate of f64 fma for ONE wave
__global__ void __launch_bounds__(64) k_indep(double* out, unsigned long long* cyc, double a, double b, int iters) {
    double x[8];
    for (int j = 0; j < 8; ++j) x[j] = a + j + threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fma(x[j], b, a);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int j = 0; j < 8; ++j) s += x[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// independent 32-bit ops issue rate
__global__ void __launch_bounds__(64) k_indep32(float* out, unsigned long long* cyc, float a, float b, int iters) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = a + j + threadIdx.x * 1e-6f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], b, a);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double* out; unsigned long long* cyc; unsigned long long h;
    CHK(hipMalloc(&out, 64 * 8)); CHK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    const char* names[] = {"dep fma f64", "dep add f64", "dep add+max f64", "readlane x2 + fma", "dep mul f64", "cvt f64->f32 fma ->f64", "dep div f64", "threshold chain (sub,max,bfi,fma)"};
#define RUN(M) k<M><<<1, 64>>>(out, cyc, 1.0, 1.0000001, 10); k<M><<<1, 64>>>(out, cyc, 1.0, 1.0000001, iters); CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); printf("%-36s %.1f cycles per iteration-unit\n", names[M], double(h) / (iters * 16.0));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    k_indep<<<1, 64>>>(out, cyc, 1.0, 1.0000001, iters); CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-36s %.2f cycles per instruction\n", "8 independent fma f64 (1 wave)", double(h) / (iters * 16.0));
    k_indep32<<<1, 64>>>((float*)out, cyc, 1.0f, 1.0000001f, iters); CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-36s %.2f cycles per instruction\n", "8 independent fma f32 (1 wave)", double(h) / (iters * 16.0));
    return 0;
}
