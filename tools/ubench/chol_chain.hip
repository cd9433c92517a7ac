// The persistent form of the blocked Cholesky (k_chol_chain: ONE launch per factorisation, tasks off a counter) against the
// launch-per-step form (k_chol_step) on the same matrices, right-hand sides riding along: U, Lt, TI and Y = U^-T R must come out
// BIT FOR BIT the same (the same arithmetic on every tile in the same order); then the time of the persistent form alone on the
// chip by grid size W and lazy period L.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../channel-pruning_amd/csrc chol_chain.hip -o chol_chain
#include "../../channel-pruning_amd/csrc/chol_step.hip"

#include <algorithm>
#include <cmath>

int cp_arena_reserve(cp_ctx *, size_t) { return CP_ERR_NOMEM; }
int cp_knob(int) { return 0; }
int cp_set_error(cp_ctx *, int code, const char *fmt, ...) {
    fprintf(stderr, "cp_set_error %d: %s\n", code, fmt);
    return code;
}

__global__ void k_fill(double *G, int p) {
    const size_t n = size_t(p) * p;
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
        const int i = int(e / p), j = int(e % p);
        G[e] = i == j ? 2.0 * p : sin(1e-3 * double(i + 1) * double(j + 1));
    }
}
__global__ void k_fill_rhs(double *R, int p, int n) {
    const size_t cnt = size_t(p) * n;
    for (size_t e = blockIdx.x * size_t(blockDim.x) + threadIdx.x; e < cnt; e += size_t(gridDim.x) * blockDim.x)
        R[e] = cos(7e-4 * double(e % 9973) + 1e-2 * double(e / n));
}
__global__ void k_diag(const double *G, int p, double *dg0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) dg0[i] = G[size_t(i) * p + i];
}

struct Bufs {
    double *G, *U, *Lt, *TI, *TIT, *dg0, *R;
    int *info;
    int nblk, p, n_pad, ninfo;
};

static Bufs make(int nblk, int n_pad) {
    Bufs b;
    b.nblk = nblk;
    b.p = nblk * NB;
    b.n_pad = n_pad;
    const size_t pp = size_t(b.p) * b.p * 8;
    hipMalloc(&b.G, pp);
    hipMalloc(&b.U, pp);
    hipMalloc(&b.Lt, pp);
    hipMalloc(&b.TI, size_t(nblk) * NB * NB * 8);
    hipMalloc(&b.TIT, size_t(nblk) * NB * NB * 8);
    hipMalloc(&b.dg0, size_t(b.p) * 8);
    hipMalloc(&b.R, size_t(b.p) * std::max(n_pad, 1) * 8);
    b.ninfo = cp_chol_info_count(nblk);
    hipMalloc(&b.info, size_t(b.ninfo) * 4);
    return b;
}
static void reset(const Bufs &b) {
    k_fill<<<2048, 256>>>(b.G, b.p);
    if (b.n_pad) k_fill_rhs<<<1024, 256>>>(b.R, b.p, b.n_pad);
    k_diag<<<(b.p + 255) / 256, 256>>>(b.G, b.p, b.dg0);
    hipMemset(b.U, 0, size_t(b.p) * b.p * 8);
    hipMemset(b.Lt, 0, size_t(b.p) * b.p * 8);
    hipMemset(b.info, 0, size_t(b.ninfo) * 4);
    hipDeviceSynchronize();
}
static float run_steps(const Bufs &b) {
    const size_t lds = size_t(LDS_DOUBLES) * sizeof(double);
    const int ntr = b.n_pad / NB;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int s = 0; s < b.nblk; ++s) {
        const int n = b.nblk - s;
        int tiles = n + ntr;
        if (s >= 2 && !(s & 1)) tiles = n * (n + 1) / 2 + n * ntr;
        else if ((s & 1) && n > 1) tiles += (n - 1) + ntr;
        k_chol_step<<<tiles, PT, lds>>>(b.G, b.U, b.Lt, b.p, b.nblk, s, b.dg0, 1e-12, b.TI, b.TIT, b.info, ntr ? b.R : nullptr, b.n_pad,
                                        ntr, 1 << 26);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
static float run_chain(const Bufs &b, int L, int W, int *total_out = nullptr) {
    const size_t lds = size_t(LDS_DOUBLES) * sizeof(double);
    const int ntr = b.n_pad / NB;
    const ChainShape sh{b.nblk, ntr, L};
    const int total = sh.total();
    if (total_out) *total_out = total;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k_chol_chain<<<std::min(W, total), PT, lds>>>(b.G, b.U, b.Lt, b.p, b.nblk, L, 0, total, 0, b.dg0, 1e-12, b.TI, b.TIT, b.info,
                                                  b.info + cp_chol_ctl_offset(b.nblk), ntr ? b.R : nullptr, b.n_pad, ntr, 1 << 24);
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) {
        printf("launch failed: %s\n", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
struct Snap {
    std::vector<double> U, Lt, TI, R;
    int info0;
};
static Snap snap(const Bufs &b) {
    Snap s;
    s.U.resize(size_t(b.p) * b.p);
    s.Lt.resize(size_t(b.p) * b.p);
    s.TI.resize(size_t(b.nblk) * NB * NB);
    s.R.resize(size_t(b.p) * std::max(b.n_pad, 1));
    hipMemcpy(s.U.data(), b.U, s.U.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(s.Lt.data(), b.Lt, s.Lt.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(s.TI.data(), b.TI, s.TI.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(s.R.data(), b.R, s.R.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&s.info0, b.info, 4, hipMemcpyDeviceToHost);
    return s;
}
static bool same(const Snap &a, const Snap &b) {
    return a.info0 == b.info0 && !memcmp(a.U.data(), b.U.data(), a.U.size() * 8) && !memcmp(a.Lt.data(), b.Lt.data(), a.Lt.size() * 8) &&
           !memcmp(a.TI.data(), b.TI.data(), a.TI.size() * 8) && !memcmp(a.R.data(), b.R.data(), a.R.size() * 8);
}

int main(int argc, char **argv) {
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
#ifdef CP_CHAIN_STATS
    if (argc > 1 && !strcmp(argv[1], "stats")) {   // where the resident workgroups' time goes: waiting for inputs / working, by task kind
        if (lds_opt_in(0) != hipSuccess) return 1;
        Bufs b1 = make(36, 512);
        for (int W : {108, 216, 288}) {
            reset(b1);
            run_chain(b1, 4, W);
            unsigned long long z[12] = {};
            hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stats), z, sizeof(z));
            reset(b1);
            const float ms = run_chain(b1, 4, W);
            unsigned long long st[12];
            hipMemcpyFromSymbol(st, HIP_SYMBOL(g_chain_stats), sizeof(st));
            const double tot = double(W) * ms * 1e-3 * 2.4e9;   // workgroup-cycles of the launch at 2.4 GHz (upper bound: not all resident at once)
            printf("W = %d: %.3f ms; workgroup-cycles %.3g\n", W, ms, tot);
            const char *nm[3] = {"pre", "chain", "rest"};
            for (int k = 0; k < 3; ++k)
                printf("  %-5s tasks %6llu  waited %.3g cycles (%.1f %% of the launch's workgroup-cycles, %.0f per task)  worked %.3g (%.1f %%, %.0f per task)\n",
                       nm[k], st[4 + k], double(st[k]), 100.0 * double(st[k]) / tot, st[4 + k] ? double(st[k]) / double(st[4 + k]) : 0.0,
                       double(st[8 + k]), 100.0 * double(st[8 + k]) / tot, st[4 + k] ? double(st[8 + k]) / double(st[4 + k]) : 0.0);
        }
        return 0;
    }
#endif
    if (lds_opt_in(0) != hipSuccess) return 1;
    int bad = 0;
    printf("## bit-for-bit against the launch-per-step form\n\n| blocks | rhs columns | L | W | tasks | identical | info[0] |\n|---|---|---|---|---|---|---|\n");
    const int shapes[][2] = {{1, 0}, {1, 128}, {2, 256}, {3, 0}, {5, 128}, {9, 512}, {14, 256}, {36, 512}};
    for (auto &sh : shapes) {
        if (quick && sh[0] > 14) continue;
        Bufs b = make(sh[0], sh[1]);
        reset(b);
        run_steps(b);
        const Snap ref = snap(b);
        for (int L = 1; L <= CHAIN_L_MAX; ++L)
            for (int W : {1, 7, 64, 400}) {
                if (sh[0] > 14 && (W == 1 || (L != 2 && L != 4))) continue;
                reset(b);
                int total = 0;
                const float ms = run_chain(b, L, W, &total);
                const Snap got = snap(b);
                const bool ok = ms >= 0 && same(ref, got);
                bad += !ok;
                printf("| %d | %d | %d | %d | %d | %s | %d |\n", sh[0], sh[1], L, W, total, ok ? "yes" : "NO", got.info0);
            }
        hipFree(b.G); hipFree(b.U); hipFree(b.Lt); hipFree(b.TI); hipFree(b.TIT); hipFree(b.dg0); hipFree(b.R); hipFree(b.info);
    }
    printf("\n%s\n\n", bad ? "MISMATCH" : "all identical");
    if (quick) return bad != 0;
    printf("## alone on the chip, 36 blocks (P = 4608), 4 right-hand-side tile columns (n = 512): ms per factorisation (best of 3)\n\n");
    Bufs b = make(36, 512);
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        reset(b);
        best = std::min(best, run_steps(b));
    }
    printf("launch per step (36 launches): %.3f ms\n\n| L \\ W |", best);
    const int Ws[] = {36, 72, 108, 144, 216, 288, 400, 512};
    for (int W : Ws) printf(" %d |", W);
    printf("\n|---|");
    for (int W : Ws) (void)W, printf("---|");
    printf("\n");
    for (int L = 1; L <= CHAIN_L_MAX; ++L) {
        printf("| %d |", L);
        for (int W : Ws) {
            float m = 1e9;
            for (int r = 0; r < 3; ++r) {
                reset(b);
                m = std::min(m, run_chain(b, L, W));
            }
            printf(" %.3f |", m);
        }
        printf("\n");
    }
    // five factorisations side by side on five streams (the refit phase of the vgg16 job has five 512-channel layers)
    printf("\n## five factorisations side by side (five streams), 36 blocks + 4 rhs columns each: ms until the last one ends\n\n");
    {
        Bufs bs[5];
        hipStream_t st[5];
        for (int q = 0; q < 5; ++q) {
            bs[q] = make(36, 512);
            hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking);
        }
        const size_t lds = size_t(LDS_DOUBLES) * sizeof(double);
        const int ntr = 4;
        auto wall = [&](auto launch) {
            float m = 1e9;
            for (int r = 0; r < 3; ++r) {
                for (int q = 0; q < 5; ++q) reset(bs[q]);
                hipEvent_t e0, e1[5];
                hipEventCreate(&e0);
                hipEventRecord(e0, st[0]);
                for (int q = 1; q < 5; ++q) hipStreamWaitEvent(st[q], e0, 0);
                for (int q = 0; q < 5; ++q) launch(bs[q], st[q]);
                float worst = 0;
                for (int q = 0; q < 5; ++q) {
                    hipEventCreate(&e1[q]);
                    hipEventRecord(e1[q], st[q]);
                }
                for (int q = 0; q < 5; ++q) {
                    hipEventSynchronize(e1[q]);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1[q]);
                    worst = std::max(worst, ms);
                }
                m = std::min(m, worst);
            }
            return m;
        };
        const float t_steps = wall([&](const Bufs &b_, hipStream_t s_) {
            for (int s = 0; s < b_.nblk; ++s) {
                const int n = b_.nblk - s;
                int tiles = n + ntr;
                if (s >= 2 && !(s & 1)) tiles = n * (n + 1) / 2 + n * ntr;
                else if ((s & 1) && n > 1) tiles += (n - 1) + ntr;
                k_chol_step<<<tiles, PT, lds, s_>>>(b_.G, b_.U, b_.Lt, b_.p, b_.nblk, s, b_.dg0, 1e-12, b_.TI, b_.TIT, b_.info, b_.R, b_.n_pad,
                                                    ntr, 1 << 26);
            }
        });
        printf("launch per step: %.3f ms\n\n| L \\ W per factorisation |", t_steps);
        const int W5[] = {36, 72, 100, 128, 160, 256};
        for (int W : W5) printf(" %d |", W);
        printf("\n|---|");
        for (int W : W5) (void)W, printf("---|");
        printf("\n");
        for (int L : {2, 4}) {
            printf("| %d |", L);
            for (int W : W5) {
                const ChainShape sh{36, ntr, L};
                const int total = sh.total();
                const float t = wall([&](const Bufs &b_, hipStream_t s_) {
                    k_chol_chain<<<W, PT, lds, s_>>>(b_.G, b_.U, b_.Lt, b_.p, b_.nblk, L, 0, total, 0, b_.dg0, 1e-12, b_.TI, b_.TIT, b_.info,
                                                     b_.info + cp_chol_ctl_offset(b_.nblk), b_.R, b_.n_pad, ntr, 1 << 24);
                });
                printf(" %.3f |", t);
            }
            printf("\n");
        }
    }
    return bad != 0;
}
