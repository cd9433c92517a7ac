// micro-model of one blocked CD step (recip+delta): which instruction groups cost what, one wave
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, double alpha, int iters, const double* rows) {
    const int lane = threadIdx.x;
    double H[4] = {1.0 + lane, 2.0, 3.0, 4.0};
    double Hs = 0.5 + lane * 1e-3, wo = 0.1 * lane, q = 3.0 + lane, Qd = 2.0, den = 0.5, keep = 0;
    double r0 = rows[lane], r1 = rows[64 + lane], r2 = rows[128 + lane], r3 = rows[192 + lane], qc = rows[256 + lane];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int la = (it & 7) * 8 + a;
            const double Hp = __builtin_fma(-wo, Qd, Hs);
            const double tmp = q - Hp;
            const double thr = __builtin_copysign(__builtin_fmax(__builtin_fabs(tmp) - alpha, 0.0), tmp);
            const double wn = thr * den;
            if (MODE != 2) keep = lane == la ? wn : keep;
            double d = wn - wo;
            double da = MODE == 1 ? d : rl(d, la);          // MODE 1: no readlane (vector value used directly)
            Hs = __builtin_fma(da, qc, Hs);
            if (MODE != 3) {
                H[0] = __builtin_fma(da, r0, H[0]); H[1] = __builtin_fma(da, r1, H[1]);
                H[2] = __builtin_fma(da, r2, H[2]); H[3] = __builtin_fma(da, r3, H[3]);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = H[0] + H[1] + H[2] + H[3] + Hs + keep;
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    double *out, *rows; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&rows, 320 * 8); hipMemset(rows, 0, 320 * 8);
    const int iters = 4000;
    const char* names[] = {"full step", "no readlane", "no latch", "no H fmas"};
#define RUN(M) k<M><<<1, 64>>>(out, cyc, 0.25, 10, rows); k<M><<<1, 64>>>(out, cyc, 0.25, iters, rows); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-14s %.1f cycles/step\n", names[M], double(h) / (iters * 8.0));
    RUN(0) RUN(1) RUN(2) RUN(3)
    return 0;
}
