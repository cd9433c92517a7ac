// Side-car probe: run NEXT TO a workload of another process on the same GPU (e.g. `./sidecar 20 & python bench.py ...`).
//   (1) one resident wave samples (s_memrealtime [100 MHz], s_memtime [shader clock]) every ~100 us for the whole run:
//       the shader clock the chip holds while the workload runs, per 0.5 s window;
//   (2) the host launches, one at a time, probe kernels of three resource footprints and times launch -> completion:
//         tiny : 64 threads, no LDS                         (a flag / iota kernel)
//         mid  : 512 threads, 80 KB LDS, <= 128 VGPRs       (fits next to ONE 72 KB / 8-wave GEMM workgroup on a CU)
//         big  : 512 threads, 158 KB LDS                    (needs a CU with no GEMM workgroup on it: today's k_potrf)
//       i.e. how long a dependent chain's next kernel waits for a place on a busy chip, by footprint.
// usage: sidecar <seconds> [period_us]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Sample {
    unsigned long long real, cyc;
};

__global__ void __launch_bounds__(64) k_monitor(Sample *buf, int n, unsigned long long period_ticks) {
    unsigned long long next = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        unsigned long long r;
        int guard = 0;
        do {
            __builtin_amdgcn_s_sleep(32);
            r = __builtin_amdgcn_s_memrealtime();
        } while (r < next && ++guard < 20000);   // bounded: a counter that does not advance must not hang the box
        const unsigned long long c = __builtin_readcyclecounter();
        if (threadIdx.x == 0) {
            buf[i].real = r;
            buf[i].cyc = c;
        }
        next += period_ticks;
    }
}

__global__ void __launch_bounds__(64) k_tiny(int *p) {
    if (threadIdx.x == 0) p[0] += 1;
}

__global__ void __launch_bounds__(512, 4) k_lds(int *p) {   // <= 128 VGPRs; dynamic LDS decides the footprint
    extern __shared__ double sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) p[0] += int(sm[5]);
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const int period_us = argc > 2 ? atoi(argv[2]) : 100;
    const int n = int(seconds * 1e6 / period_us);
    Sample *buf;
    hipMalloc(&buf, sizeof(Sample) * size_t(n));
    hipMemset(buf, 0, sizeof(Sample) * size_t(n));
    int *p;
    hipMalloc(&p, 64);
    hipMemset(p, 0, 64);
    hipStream_t sm_, sp;
    hipStreamCreateWithFlags(&sm_, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sp, hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
    // warm the three probe kernels
    k_tiny<<<1, 64, 0, sp>>>(p);
    k_lds<<<1, 512, 80 * 1024, sp>>>(p);
    k_lds<<<1, 512, 158 * 1024, sp>>>(p);
    hipStreamSynchronize(sp);
    k_monitor<<<1, 64, 0, sm_>>>(buf, n, (unsigned long long)(period_us) * 100ull);
    const double t_start = now_us();
    printf("start_epoch %.3f\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count());
    struct Rec {
        double t, lat;
        int cls;
    };
    std::vector<Rec> recs;
    int cls = 0;
    while (now_us() - t_start < seconds * 1e6 - 2e5) {
        const double t0 = now_us();
        if (cls == 0) k_tiny<<<1, 64, 0, sp>>>(p);
        else k_lds<<<1, 512, (cls == 1 ? 80 : 158) * 1024, sp>>>(p);
        hipStreamSynchronize(sp);
        const double t1 = now_us();
        recs.push_back({t0 - t_start, t1 - t0, cls});
        cls = (cls + 1) % 3;
        while (now_us() - t1 < 150.0) {}   // ~5 k probes per second in total
    }
    hipStreamSynchronize(sm_);
    std::vector<Sample> h(n);
    hipMemcpy(h.data(), buf, sizeof(Sample) * size_t(n), hipMemcpyDeviceToHost);
    // per 0.5 s window: shader clock (median / min of the per-sample estimates), probe latency medians and p90 by class
    const double win = 0.5e6;
    const int nwin = int(seconds * 1e6 / win);
    printf("| window s | shader clock GHz median | min | max | tiny us med / p90 | mid (80 KB LDS) us med / p90 | big (158 KB LDS) us med / p90 |\n");
    printf("|---|---|---|---|---|---|---|\n");
    const unsigned long long r_first = h[0].real;
    for (int w = 0; w < nwin; ++w) {
        std::vector<double> ghz;
        for (int i = 1; i < n; ++i) {
            if (!h[i].real || !h[i - 1].real) continue;
            const double t = double(h[i].real - r_first) / 100.0;   // us since the first sample
            if (t < w * win || t >= (w + 1) * win) continue;
            ghz.push_back(double(h[i].cyc - h[i - 1].cyc) / double(h[i].real - h[i - 1].real) * 0.1);
        }
        std::vector<double> lat[3];
        for (const Rec &r : recs)
            if (r.t >= w * win && r.t < (w + 1) * win) lat[r.cls].push_back(r.lat);
        auto pct = [](std::vector<double> &v, double q) {
            if (v.empty()) return 0.0;
            std::sort(v.begin(), v.end());
            return v[std::min(v.size() - 1, size_t(q * v.size()))];
        };
        printf("| %.1f | %.3f | %.3f | %.3f | %.0f / %.0f | %.0f / %.0f | %.0f / %.0f |\n", w * 0.5, pct(ghz, 0.5), pct(ghz, 0.0),
               pct(ghz, 0.999), pct(lat[0], 0.5), pct(lat[0], 0.9), pct(lat[1], 0.5), pct(lat[1], 0.9), pct(lat[2], 0.5),
               pct(lat[2], 0.9));
    }
    return 0;
}
