// What the f64 MFMA pipe delivers and at which shader clock it does so.
//   v_mfma_f64_16x16x4_f64 issued back to back from W waves per SIMD on every CU; every wave stamps the shader-clock
//   counter (s_memtime) and the constant 100 MHz counter (s_memrealtime) around its loop:
//       effective shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
//       cycles per MFMA and SIMD = d(s_memtime) x 4 SIMDs x CUs / (instructions of the launch)
//   short launches (about 1 ms: the length of a Gram GEMM of the pruning job) and long ones (about 0.5 s: what the power
//   management settles at).
// Build / run on the GPU box: hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock && ./mfma_clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double v4 __attribute__((ext_vector_type(4)));

struct Stamp {
    unsigned long long cyc, real;
};

// MINW = 1: a whole SIMD's register file for one wave -- the compiler then keeps the accumulators in AccVGPRs; MINW = 2: a
// budget of 256 registers -- VGPR accumulators, the form the kernels of the library use.  Same instruction, other issue rate.
template <int NACC, int MINW>
__global__ void __launch_bounds__(256, MINW) k_mfma(double *out, Stamp *st, int iters, double a0) {
    v4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4{0., 0., 0., 0.};
    double a = a0 + threadIdx.x * 1e-3, b = a0 - threadIdx.x * 2e-3;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        st[w].cyc = t1 - t0;
        st[w].real = r1 - r0;
    }
}

// the same stamps around a plain f64 FMA loop (VALU, no MFMA): the clock the chip holds without the matrix pipe
__global__ void __launch_bounds__(256) k_fma(double *out, Stamp *st, int iters, double a0) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-3;
    const double m = 1.0000001, c = 1e-9;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        st[w].cyc = t1 - t0;
        st[w].real = r1 - r0;
    }
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    double *out;
    Stamp *st;
    const int max_waves = cus * 4 * 8;
    hipMalloc(&out, 64);
    hipMalloc(&st, sizeof(Stamp) * max_waves);
    std::vector<Stamp> h(max_waves);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("| kernel | waves/SIMD | CUs | launch ms | TFLOP/s | shader clock GHz (median wave) | min..max GHz | cycles per MFMA and SIMD |\n");
    printf("|---|---|---|---|---|---|---|---|\n");
    auto report = [&](const char *name, int wps, int used_cus, int waves, double instr_per_wave, float ms, bool mfma) {
        hipMemcpy(h.data(), st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
        std::vector<double> ghz(waves), cyc(waves);
        for (int i = 0; i < waves; ++i) {
            ghz[i] = double(h[i].cyc) / double(h[i].real) * 0.1;
            cyc[i] = double(h[i].cyc);
        }
        std::sort(ghz.begin(), ghz.end());
        std::sort(cyc.begin(), cyc.end());
        const double tf = mfma ? double(waves) * instr_per_wave * 2048.0 / (ms * 1e-3) / 1e12
                               : double(waves) * instr_per_wave * 128.0 / (ms * 1e-3) / 1e12;
        // a SIMD runs wps waves: instructions per SIMD = wps x instr_per_wave over the median wave's cycles
        const double cpi = cyc[waves / 2] / (double(wps) * instr_per_wave);
        printf("| %s | %d | %d | %.3f | %.2f | %.3f | %.3f..%.3f | %.1f |\n", name, wps, used_cus, ms, tf, ghz[waves / 2], ghz[0],
               ghz[waves - 1], cpi);
    };
    for (int form = 0; form < 2; ++form)
    for (int lng = 0; lng < 2; ++lng) {
        for (int wps : {1, 2, 4, 8}) {
            for (int used : {1, cus}) {
                if (used == 1 && lng) continue;
                // 256 threads = 4 waves = one per SIMD; wps workgroups per CU
                const int blocks = used * wps;
                const int waves = blocks * 4;
                const int iters = (lng ? 400000 : 1000) / wps;
                if (form == 0 && wps > 2) continue;   // the AccVGPR build holds 2 waves per SIMD at most
                auto kern = form ? k_mfma<8, 2> : k_mfma<8, 1>;
                kern<<<blocks, 256>>>(out, st, 50, 0.5);
                hipEventRecord(e0);
                kern<<<blocks, 256>>>(out, st, iters, 0.5);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                report(form ? (lng ? "mfma_f64 VGPR acc, long" : "mfma_f64 VGPR acc, short")
                            : (lng ? "mfma_f64 AccVGPR acc, long" : "mfma_f64 AccVGPR acc, short"),
                       wps, used, waves, double(iters) * 8.0, ms, true);
            }
        }
    }
    for (int wps : {1, 4}) {
        const int blocks = cus * wps, waves = blocks * 4, iters = 400000 / wps;
        k_fma<<<blocks, 256>>>(out, st, 50, 0.5);
        hipEventRecord(e0);
        k_fma<<<blocks, 256>>>(out, st, iters, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        report("fma_f64 VALU", wps, cus, waves, double(iters) * 8.0, ms, false);
    }
    // one wave alone, dependent chain of MFMAs on ONE accumulator: the latency of the instruction
    {
        k_mfma<1, 2><<<1, 64>>>(out, st, 50, 0.5);
        hipEventRecord(e0);
        k_mfma<1, 2><<<1, 64>>>(out, st, 20000, 0.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), st, sizeof(Stamp), hipMemcpyDeviceToHost);
        printf("one wave, one VGPR accumulator (dependent chain): %.1f cycles per MFMA at %.3f GHz\n", double(h[0].cyc) / 20000.0,
               double(h[0].cyc) / double(h[0].real) * 0.1);
    }
    return 0;
}
