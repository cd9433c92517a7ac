// The factorisation launches of csrc/chol_step.hip ALONE on a 4608 x 4608 matrix: per launch the time and the workgroup
// count, and for the workgroups of one bulk launch (every tile below the block row: 256-deep updates) the shader-clock
// stamps of their phases -- tile load | update loop | store.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../channel-pruning_amd/csrc chol_bulk.hip -o chol_bulk
#define CP_CHOL_STAMPS
#include "../../channel-pruning_amd/csrc/chol_step.hip"

#include <algorithm>
#include <cmath>

int cp_arena_reserve(cp_ctx *, size_t) { return CP_ERR_NOMEM; }   // only cp_debug_lds_hog wants it (not used here)
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
__global__ void k_diag(const double *G, int p, double *dg0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) dg0[i] = G[size_t(i) * p + i];
}

int main(int argc, char **argv) {
    (void)argc, (void)argv;
    const int nblk = 36, p = nblk * NB;
    double *G, *U, *Lt, *TI, *TIT, *dg0;
    int *info;
    hipMalloc(&G, size_t(p) * p * 8);
    hipMalloc(&U, size_t(p) * p * 8);
    hipMalloc(&Lt, size_t(p) * p * 8);
    hipMalloc(&TI, size_t(nblk) * NB * NB * 8);
    hipMalloc(&TIT, size_t(nblk) * NB * NB * 8);
    hipMalloc(&dg0, size_t(p) * 8);
    hipMalloc(&info, 4096);
    hipMemset(U, 0, size_t(p) * p * 8);
    hipMemset(Lt, 0, size_t(p) * p * 8);
    if (lds_opt_in(0) != hipSuccess) return 1;
    const size_t lds = size_t(LDS_DOUBLES) * sizeof(double);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<unsigned long long> st(4096 * 8), st5(4096 * 8);
    for (int pass = 0; pass < 2; ++pass) {
        k_fill<<<2048, 256>>>(G, p);
        k_diag<<<(p + 255) / 256, 256>>>(G, p, dg0);
        hipMemset(info, 0, 4096);
        hipDeviceSynchronize();
        float total = 0;
        if (pass) printf("| step | workgroups | us |\n|---|---|---|\n");
        for (int s = 0; s < nblk; ++s) {
            const int n = nblk - s;
            int tiles = n;
            if (s >= 2 && !(s & 1)) tiles = n * (n + 1) / 2;
            else if ((s & 1) && n > 1) tiles += n - 1;
            hipEventRecord(e0);
            k_chol_step<<<tiles, PT, lds>>>(G, U, Lt, p, nblk, s, dg0, 1e-12, TI, TIT, info, nullptr, 0, 0, 1 << 26);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            total += ms;
            if (pass && (s < 12 || s % 4 == 0)) printf("| %d | %d | %.1f |\n", s, tiles, ms * 1e3);
            if (pass && s == 4) hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(cp_chol_stamps), st.size() * 8);
            if (pass && s == 5) hipMemcpyFromSymbol(st5.data(), HIP_SYMBOL(cp_chol_stamps), st5.size() * 8);
        }
        int h[4];
        hipMemcpy(h, info, 16, hipMemcpyDeviceToHost);
        if (pass) printf("\nall %d steps: %.3f ms (info[0] = %d)\n", nblk, total, h[0]);
    }
    {   // correctness of the factor this run left: U^T U against the matrix k_fill builds, U strictly upper-zero below the
        // diagonal, TI_b U_bb = I and the operator's diagonal slots (sampled; host arithmetic)
        std::vector<double> hU(size_t(p) * p), hTI(size_t(nblk) * NB * NB);
        hipMemcpy(hU.data(), U, hU.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(hTI.data(), TI, hTI.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0, lower = 0, inv_err = 0;
        unsigned rs = 12345u;
        auto rnd = [&](int m) { rs = rs * 1664525u + 1013904223u; return int((rs >> 8) % unsigned(m)); };
        for (int t = 0; t < 3000; ++t) {
            int i = rnd(p), j = rnd(p);
            if (t < 600) j = std::min(p - 1, i + rnd(40));          // near the diagonal, inside diagonal tiles
            if (i > j) std::swap(i, j);
            double acc = 0;
            for (int k = 0; k <= i; ++k) acc += hU[size_t(k) * p + i] * hU[size_t(k) * p + j];
            const double want = i == j ? 2.0 * p : sin(1e-3 * double(i + 1) * double(j + 1));
            worst = std::max(worst, fabs(acc - want));
            if (i != j && (i / NB) == (j / NB)) lower = std::max(lower, fabs(hU[size_t(j) * p + i]));   // inside a diagonal tile, below the diagonal
        }
        for (int b = 0; b < nblk; b += 7)
            for (int t = 0; t < 400; ++t) {
                const int i = rnd(NB), j = rnd(NB);
                double acc = 0;
                for (int k = 0; k < NB; ++k) acc += hTI[size_t(b) * NB * NB + size_t(i) * NB + k] * hU[size_t(b * NB + k) * p + b * NB + j];
                inv_err = std::max(inv_err, fabs(acc - (i == j ? 1.0 : 0.0)));
            }
        printf("\ncheck: max |U^T U - G| = %.3e (|G_ii| = %.0f), max |U| below the diagonal inside diagonal tiles = %.3e, "
               "max |TI_b U_bb - I| = %.3e\n", worst, 2.0 * p, lower, inv_err);
    }
    // step 4: workgroups [0, 32) are block row 4 (diagonal + panels), the rest bulk tiles with K = 256
    const int n4 = nblk - 4, tiles4 = n4 * (n4 + 1) / 2;
    std::vector<double> load, upd, store, all;
    unsigned long long first = ~0ull, last = 0;
    for (int w = n4; w < tiles4; ++w) {
        const unsigned long long *q = &st[size_t(w) * 8];
        if (!q[0] || !q[3]) continue;
        load.push_back(double(q[1] - q[0]));
        upd.push_back(double(q[2] - q[1]));
        store.push_back(double(q[3] - q[2]));
        all.push_back(double(q[3] - q[0]));
        first = std::min(first, q[0]);
        last = std::max(last, q[3]);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    printf("\nstep 4, %zu bulk workgroups (256-deep update of a 128 x 128 tile), shader cycles (s_memtime, ~2.39 GHz): median / max\n", all.size());
    printf("| tile load | update loop (16 chunks) | store | whole |\n|---|---|---|---|\n");
    printf("| %.0f / %.0f | %.0f / %.0f | %.0f / %.0f | %.0f / %.0f |\n", med(load), mx(load), med(upd), mx(upd), med(store),
           mx(store), med(all), mx(all));
    {   // the look-ahead factorisation of the LAST step's diagonal tile, panel by panel
        std::vector<unsigned long long> ds(128);
        hipMemcpyFromSymbol(ds.data(), HIP_SYMBOL(cp_chol_diag_stamps), ds.size() * 8);
        printf("\nlook-ahead factorisation (cycles): first 16 x 16 block in registers %llu\n", ds[1] - ds[0]);
        printf("| panel | phase A: U_pj = T_p^T A_pj (wave 0) | barrier | wave 0: update of block p+1 | wave 0: factorisation of block p+1 | wave 1: its trailing blocks | whole panel |\n|---|---|---|---|---|---|---|\n");
        for (int q = 0; q < 7; ++q) {
            const unsigned long long *d = &ds[8 + 8 * q];
            printf("| %d | %llu | %llu | %llu | %llu | %llu | %llu |\n", q, d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[2], d[6] - d[0]);
        }
    }
    // step 5 (odd: block row 5 + block row 6 riding along, no other bulk): the serial piece, phase by phase
    {
        const unsigned long long *d = &st5[0];
        printf("\nstep 5, diagonal workgroup (cycles): tile load %llu | K=128 update %llu | to LDS %llu | 128x128 factorisation %llu | "
               "output + flag %llu | inverse (off the chain) %llu\n", d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4],
               d[6] - d[5]);
        std::vector<double> upd, wait, opl, solve;   // (stamps of different CUs are not comparable: no cross-workgroup column)
        const int n5 = nblk - 5;
        for (int w = 1; w < n5; ++w) {
            const unsigned long long *q = &st5[size_t(w) * 8];
            if (!q[0] || !q[5]) continue;
            upd.push_back(double(q[2] - q[0]));
            wait.push_back(double(q[3] - q[2]));
            opl.push_back(double(q[4] - q[3]));
            solve.push_back(double(q[5] - q[4]));
        }
        printf("step 5, %zu panel workgroups (median / max cycles): load + update %.0f / %.0f | wait for the flag %.0f / %.0f | operator "
               "load %.0f / %.0f | substitution + stores %.0f / %.0f\n",
               upd.size(), med(upd), mx(upd), med(wait), mx(wait), med(opl), mx(opl), med(solve), mx(solve));
    }
    return 0;
}
