// aggregate kernel dispatch rate: T host threads, one stream each, N dependent tiny kernels per stream
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_tiny(int *p) {
    if (threadIdx.x == 0 && p) p[blockIdx.x] += 1;
}
__global__ void k_spin(int *p, int cycles) {  // ~cycles of work on 1 workgroup
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && p) p[blockIdx.x] += 1;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2000;
    const int B = argc > 2 ? atoi(argv[2]) : 0;  // background streams, each running back-to-back 4 ms one-wave kernels
    std::vector<hipStream_t> bg(B);
    std::vector<std::thread> bgth;
    volatile bool stop = false;
    int *bgbuf = nullptr;
    hipMalloc(&bgbuf, 4096);
    for (int i = 0; i < B; ++i) {
        hipStreamCreateWithFlags(&bg[i], hipStreamNonBlocking);
        bgth.emplace_back([&, i]() {
            while (!stop) {
                k_spin<<<1, 64, 0, bg[i]>>>(bgbuf, 9600000);
                hipStreamSynchronize(bg[i]);
            }
        });
    }
    for (int spin : {0, 20000}) {
        for (int T : {1, 2, 3, 4, 5, 6, 8, 12, 18}) {
            std::vector<hipStream_t> st(T);
            std::vector<int *> buf(T);
            for (int i = 0; i < T; ++i) {
                hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
                hipMalloc(&buf[i], 4096);
                hipMemset(buf[i], 0, 4096);
            }
            hipDeviceSynchronize();
            auto run = [&](int i, int n) {
                for (int k = 0; k < n; ++k) {
                    if (spin) k_spin<<<1, 64, 0, st[i]>>>(buf[i], spin);
                    else k_tiny<<<1, 64, 0, st[i]>>>(buf[i]);
                }
                hipStreamSynchronize(st[i]);
            };
            {  // warm
                std::vector<std::thread> th;
                for (int i = 0; i < T; ++i) th.emplace_back(run, i, 50);
                for (auto &t : th) t.join();
            }
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int i = 0; i < T; ++i) th.emplace_back(run, i, N);
            for (auto &t : th) t.join();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("spin %5d cycles  streams %2d: %8.1f k launches/s aggregate, %6.2f us per launch per stream\n", spin, T,
                   double(T) * N / ms, ms * 1e3 / N);
            for (int i = 0; i < T; ++i) {
                hipStreamDestroy(st[i]);
                hipFree(buf[i]);
            }
        }
    }
    stop = true;
    for (auto &t : bgth) t.join();
    return 0;
}
