// Do the f64 matrix pipe and the f64 vector pipe of a gfx950 SIMD add up?
//   (a) wave mix: every CU runs WM waves per SIMD of back-to-back v_mfma_f64_16x16x4_f64 next to WV waves per SIMD of
//       back-to-back v_fma_f64 (one workgroup of 256 x (WM + WV) threads per CU: wave w sits on SIMD w % 4, the first 4 WM
//       waves take the MFMA role); each role stops after its own instruction count and stamps s_memtime / s_memrealtime;
//   (b) one stream: every wave interleaves R independent v_fma_f64 after each MFMA (what a GEMM wave that computes part of
//       its tile on the vector pipe would issue).
//   Rates are taken over the window in which BOTH roles run (the shorter role's own duration), so a role that finishes
//   early does not flatter the other.
// Build / run on the GPU box: hipcc --offload-arch=gfx950 -O3 mfma_valu_mix.hip -o mfma_valu_mix && ./mfma_valu_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double v4 __attribute__((ext_vector_type(4)));

struct Stamp {
    unsigned long long cyc, real;
};

__device__ __forceinline__ void stamp_out(Stamp *st, unsigned long long t0, unsigned long long r0) {
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        st[w].cyc = t1 - t0;
        st[w].real = r1 - r0;
    }
}

// waves [0, 4 WM) of the workgroup: MFMA role; the rest: VALU role
__global__ void __launch_bounds__(1024) k_mix(double *out, Stamp *st, int wm, int it_m, int it_v, double a0) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4 * wm) {
        v4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = v4{0., 0., 0., 0.};
        const double a = a0 + threadIdx.x * 1e-3, b = a0 - threadIdx.x * 2e-3;
        for (int it = 0; it < it_m; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        stamp_out(st, t0, r0);
        if (s == 12345.678) out[0] = s;
    } else {
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x * 1e-3;
        const double m = 1.0000001, c = 1e-9;
        for (int it = 0; it < it_v; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fma(x[i], m, c);
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
        stamp_out(st, t0, r0);
        if (s == 12345.678) out[0] = s;
    }
}

// one instruction stream: R independent v_fma_f64 behind every MFMA
template <int R>
__global__ void __launch_bounds__(256) k_interleave(double *out, Stamp *st, int iters, double a0) {
    v4 acc[8];
    double x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4{0., 0., 0., 0.};
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x * 1e-3;
    const double a = a0 + threadIdx.x * 1e-3, b = a0 - threadIdx.x * 2e-3;
    const double m = 1.0000001, c = 1e-9;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < R; ++r) x[(i * R + r) & 15] = fma(x[(i * R + r) & 15], m, c);
        }
        // keep the order above in the schedule: one MFMA, then its R vector FMAs
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (R > 0) __builtin_amdgcn_sched_group_barrier(0x002, R, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    stamp_out(st, t0, r0);
    if (s == 12345.678) out[0] = s;
}

static double median_of(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n\n", prop.gcnArchName, cus);
    double *out;
    Stamp *st;
    const int max_waves = cus * 16;
    hipMalloc(&out, 64);
    hipMalloc(&st, sizeof(Stamp) * max_waves);
    std::vector<Stamp> h(max_waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);

    printf("(a) wave mix, one workgroup per CU, ~100 ms launches\n");
    printf("| MFMA waves/SIMD | VALU waves/SIMD | launch ms | MFMA TFLOP/s (own window) | VALU TFLOP/s (own window) | sum | "
           "MFMA cycles/instr/SIMD | VALU cycles/instr/SIMD | clock GHz |\n|---|---|---|---|---|---|---|---|---|\n");
    struct Mix {
        int wm, wv;
    };
    for (Mix mx : {Mix{2, 0}, Mix{0, 2}, Mix{0, 4}, Mix{1, 1}, Mix{2, 1}, Mix{2, 2}, Mix{1, 2}, Mix{1, 3}}) {
        const int wpw = 4 * (mx.wm + mx.wv);           // waves per workgroup
        // aim at ~100 ms for either role alone: MFMA 102 cycles/instr/SIMD, VALU ~5 cycles/instr/SIMD at 2.3 GHz
        const int it_m = mx.wm ? int(2.3e8 / 102.0 / 8.0 / mx.wm) : 0;
        const int it_v = mx.wv ? int(2.3e8 / 5.0 / 64.0 / mx.wv) : 0;
        k_mix<<<cus, 64 * wpw>>>(out, st, mx.wm, 50, 50, 0.5);
        hipEventRecord(e0);
        k_mix<<<cus, 64 * wpw>>>(out, st, mx.wm, it_m, it_v, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const int waves = cus * wpw;
        hipMemcpy(h.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
        std::vector<double> cm, cv, ghz, rm, rv;
        for (int i = 0; i < waves; ++i) {
            const bool is_m = (i % wpw) < 4 * mx.wm;
            (is_m ? cm : cv).push_back(double(h[i].cyc));
            (is_m ? rm : rv).push_back(double(h[i].real) * 1e-8);   // seconds (100 MHz counter)
            ghz.push_back(double(h[i].cyc) / double(h[i].real) * 0.1);
        }
        const double tm = mx.wm ? median_of(rm) : 0, tv = mx.wv ? median_of(rv) : 0;
        const double tf_m = mx.wm ? double(cus) * 4 * mx.wm * it_m * 8.0 * 2048.0 / tm / 1e12 : 0;
        const double tf_v = mx.wv ? double(cus) * 4 * mx.wv * it_v * 64.0 * 128.0 / tv / 1e12 : 0;
        const double cpm = mx.wm ? median_of(cm) / (double(mx.wm) * it_m * 8.0) : 0;
        const double cpv = mx.wv ? median_of(cv) / (double(mx.wv) * it_v * 64.0) : 0;
        printf("| %d | %d | %.1f | %.2f (%.1f ms) | %.2f (%.1f ms) | %.2f | %.1f | %.2f | %.3f |\n", mx.wm, mx.wv, ms, tf_m,
               tm * 1e3, tf_v, tv * 1e3, tf_m + tf_v, cpm, cpv, median_of(ghz));
    }

    printf("\n(b) one stream: R v_fma_f64 behind every MFMA, 2 waves per SIMD (512 workgroups of 256), ~100 ms launches\n");
    printf("| R | launch ms | MFMA TFLOP/s | VALU TFLOP/s | sum | cycles per MFMA (+R FMAs) and SIMD | clock GHz |\n|---|---|---|---|---|---|---|\n");
    auto run_b = [&](int R, auto kern) {
        const int wps = 2, blocks = cus * wps, waves = blocks * 4;
        const int iters = int(2.3e8 / (102.0 + 5.0 * R) / 8.0 / wps);
        kern<<<blocks, 256>>>(out, st, 50, 0.5);
        hipEventRecord(e0);
        kern<<<blocks, 256>>>(out, st, iters, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int i = 0; i < waves; ++i) {
            cyc.push_back(double(h[i].cyc));
            ghz.push_back(double(h[i].cyc) / double(h[i].real) * 0.1);
        }
        const double n_m = double(waves) * iters * 8.0;
        const double tf_m = n_m * 2048.0 / (ms * 1e-3) / 1e12, tf_v = n_m * R * 128.0 / (ms * 1e-3) / 1e12;
        printf("| %d | %.1f | %.2f | %.2f | %.2f | %.1f | %.3f |\n", R, ms, tf_m, tf_v, tf_m + tf_v,
               median_of(cyc) / (double(wps) * iters * 8.0), median_of(ghz));
    };
    run_b(0, k_interleave<0>);
    run_b(2, k_interleave<2>);
    run_b(4, k_interleave<4>);
    run_b(8, k_interleave<8>);
    run_b(12, k_interleave<12>);
    run_b(16, k_interleave<16>);
    run_b(24, k_interleave<24>);
    return 0;
}
