// What bounds the f64 "TN" GEMM of csrc/gemm_f64.hip?  Stand-alone variants of its main loop (full tiles, no split-K, no
// triangle), timed on tile counts that do and do not fill the chip evenly:
//     C[M,N] = sum_k A[k,m] B[k,n],   A: K x M, B: K x N (k-major), K = 4992
//   template <TMW, TNW, WR, WC, BK, PIPE, MINW>: WR x WC waves per workgroup, each a TMW x TNW tile of 16 x 16 MFMA tiles;
//   BK rows per LDS stage (two stages, one barrier per stage); PIPE = 1: the MFMA operands of k-group kk + 1 are read from
//   LDS into a second register set before the MFMAs of kk are issued; MINW = waves per SIMD the register budget is cut for.
// Build / run on the GPU box: hipcc --offload-arch=gfx950 -O3 gemm_probe.hip -o gemm_probe && ./gemm_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int LPAD = 16, SB = 8;

__device__ __forceinline__ void decode_blocked(int idx, int tiles_m, int tiles_n, int &ti, int &tj) {
    const int per_row = SB * tiles_n;
    int I = idx / per_row;
    const int nbr = (tiles_m + SB - 1) / SB;
    if (I > nbr - 1) I = nbr - 1;
    const int rem = idx - I * per_row;
    const int h = min(SB, tiles_m - I * SB);
    const int J = rem / (h * SB);
    const int r2 = rem - J * h * SB;
    const int w = min(SB, tiles_n - J * SB);
    ti = I * SB + r2 / w;
    tj = J * SB + r2 % w;
}

template <int TMW, int TNW, int WR, int WC, int BK, int PIPE, int MINW>
__global__ void __launch_bounds__(64 * WR * WC, MINW)
k_gemm(int M, int N, int K, const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb,
       double *__restrict__ C, int ldc, int tiles_m, int tiles_n) {
    constexpr int NTH = 64 * WR * WC;
    constexpr int TM = WR * TMW, TN = WC * TNW;
    constexpr int ALD = TM + LPAD, BLD = TN + LPAD;
    constexpr int FRM = TMW / 16, FRN = TNW / 16;
    constexpr int ACH = BK * TM / 2 / NTH, BCH = BK * TN / 2 / NTH;   // double2 chunks per thread and stage
    static_assert(ACH >= 1 && BCH >= 1, "stage too small for the workgroup");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *As2 = smem, *Bs2 = smem + 2 * BK * ALD;

    const int tid = threadIdx.x;
    const int n_tiles = tiles_m * tiles_n;
    const int L = blockIdx.x;
    int tile = L;
    if (n_tiles >= 64) {
        const int xcd = L & 7, slot = L >> 3;
        const int base = n_tiles >> 3, rem = n_tiles & 7;
        tile = xcd * base + (xcd < rem ? xcd : rem) + slot;
    }
    int ti, tj;
    if (n_tiles >= 64) decode_blocked(tile, tiles_m, tiles_n, ti, tj);
    else { ti = tile / tiles_n; tj = tile - ti * tiles_n; }
    const int m0 = ti * TM, n0 = tj * TN;
    const int nk = K / BK;

    v2f64 ar[ACH], br[BCH];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + NTH * i, row = c / (TM / 2), col = (c % (TM / 2)) * 2;
            ar[i] = *reinterpret_cast<const v2f64 *>(A + size_t(kt * BK + row) * lda + m0 + col);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int c = tid + NTH * i, row = c / (TN / 2), col = (c % (TN / 2)) * 2;
            br[i] = *reinterpret_cast<const v2f64 *>(B + size_t(kt * BK + row) * ldb + n0 + col);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int c = tid + NTH * i, row = c / (TM / 2), col = (c % (TM / 2)) * 2;
            *reinterpret_cast<v2f64 *>(&As2[buf * BK * ALD + row * ALD + col]) = ar[i];
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) {
            const int c = tid + NTH * i, row = c / (TN / 2), col = (c % (TN / 2)) * 2;
            *reinterpret_cast<v2f64 *>(&Bs2[buf * BK * BLD + row * BLD + col]) = br[i];
        }
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WC) * TMW, wn = (wave % WC) * TNW;
    const int fk = lane >> 4, fi = lane & 15;

    v4f64 acc[FRM][FRN];
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FRN; ++j) acc[i][j] = v4f64{0., 0., 0., 0.};

    load_tile(0);
    stage_write(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const double *As = As2 + (kt & 1) * BK * ALD, *Bs = Bs2 + (kt & 1) * BK * BLD;
        if (kt + 1 < nk) load_tile(kt + 1);
        if (PIPE) {
            double a[2][FRM], b[2][FRN];
#pragma unroll
            for (int i = 0; i < FRM; ++i) a[0][i] = As[fk * ALD + wm + i * 16 + fi];
#pragma unroll
            for (int j = 0; j < FRN; ++j) b[0][j] = Bs[fk * BLD + wn + j * 16 + fi];
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int i = 0; i < FRM; ++i) a[nxt][i] = As[((kk + 1) * 4 + fk) * ALD + wm + i * 16 + fi];
#pragma unroll
                    for (int j = 0; j < FRN; ++j) b[nxt][j] = Bs[((kk + 1) * 4 + fk) * BLD + wn + j * 16 + fi];
                }
#pragma unroll
                for (int i = 0; i < FRM; ++i)
#pragma unroll
                    for (int j = 0; j < FRN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                // keep the reads of kk + 1 ahead of the MFMAs of kk in the schedule
                __builtin_amdgcn_sched_group_barrier(0x100, FRM + FRN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, FRM * FRN, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[FRM], b[FRN];
#pragma unroll
                for (int i = 0; i < FRM; ++i) a[i] = As[(kk * 4 + fk) * ALD + wm + i * 16 + fi];
#pragma unroll
                for (int j = 0; j < FRN; ++j) b[j] = Bs[(kk * 4 + fk) * BLD + wn + j * 16 + fi];
#pragma unroll
                for (int i = 0; i < FRM; ++i)
#pragma unroll
                    for (int j = 0; j < FRN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) stage_write((kt + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FRN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + fk + 4 * r, col = n0 + wn + j * 16 + fi;
                C[size_t(row) * ldc + col] = acc[i][j][r];
            }
}

// the main loop without any memory traffic but the LDS reads: what the MFMA + ds_read pattern alone sustains
template <int FRM, int FRN, int NTH, int MINW>
__global__ void __launch_bounds__(NTH, MINW) k_lds_mfma(double *out, int iters) {
    __shared__ double s[16 * 144 * 2];
    for (int i = threadIdx.x; i < 16 * 144 * 2; i += NTH) s[i] = 1e-3 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, fk = lane >> 4, fi = lane & 15, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 16 * FRM % 128, wn = (wave & 1) * 16 * FRN % 128;
    v4f64 acc[FRM][FRN];
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FRN; ++j) acc[i][j] = v4f64{0., 0., 0., 0.};
    for (int it = 0; it < iters; ++it) {
        const double *As = s + (it & 1) * 16 * 144, *Bs = s + ((it & 1) ^ 1) * 16 * 144;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[FRM], b[FRN];
#pragma unroll
            for (int i = 0; i < FRM; ++i) a[i] = As[(kk * 4 + fk) * 144 + (wm + i * 16) % 128 + fi];
#pragma unroll
            for (int j = 0; j < FRN; ++j) b[j] = Bs[(kk * 4 + fk) * 144 + (wn + j * 16) % 128 + fi];
#pragma unroll
            for (int i = 0; i < FRM; ++i)
#pragma unroll
                for (int j = 0; j < FRN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    double sum = 0;
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FRN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 12345.678) out[0] = sum;
}

// back-to-back MFMAs on 8 accumulators held in VGPRs (ACC_AGPR = 0) or in AccVGPRs (1): same instruction, other register file
template <int ACC_AGPR>
__global__ void __launch_bounds__(512, 2) k_pure(double *out, int iters, double a0) {
    v4f64 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f64{0., 0., 0., 0.};
    const double a = a0 + threadIdx.x * 1e-3, b = a0 - threadIdx.x * 2e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (ACC_AGPR)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}

struct Shape {
    int tm, tn;
};

template <int TMW, int TNW, int WR, int WC, int BK, int PIPE, int MINW>
static void run_variant(const char *name, const double *A, const double *B, double *C, int ld, int K, hipEvent_t e0, hipEvent_t e1) {
    constexpr int TM = WR * TMW, TN = WC * TNW, NTH = 64 * WR * WC;
    const size_t lds = size_t(2) * BK * (TM + LPAD + TN + LPAD) * sizeof(double);
    auto kern = k_gemm<TMW, TNW, WR, WC, BK, PIPE, MINW>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NTH, lds);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    // tile grids in units of 128 x 128 output blocks: 256 / 512 / 595 / 1024 / 1296 blocks
    const Shape shapes[] = {{16, 16}, {16, 32}, {17, 35}, {32, 32}, {36, 36}};
    printf("| %s (%d x %d tile, %d thr, LDS %zu KB, %d regs, %d wg/CU) |", name, TM, TN, NTH, lds / 1024, fa.numRegs, occ);
    for (const Shape &sh : shapes) {
        const int M = sh.tm * 128, N = sh.tn * 128;
        if (M % TM || N % TN) {
            printf(" - |");
            continue;
        }
        const int tiles_m = M / TM, tiles_n = N / TN;
        std::vector<float> t;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            kern<<<tiles_m * tiles_n, NTH, lds>>>(M, N, K, A, ld, B, ld, C, ld, tiles_m, tiles_n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        const double ms = t[t.size() / 2];
        printf(" %.3f ms %.1f TF |", ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs\n\n", prop.gcnArchName, cus);
    const int ld = 36 * 128, K = 4992;
    double *A, *B, *C, *out;
    hipMalloc(&A, size_t(K) * ld * 8);
    hipMalloc(&B, size_t(K) * ld * 8);
    hipMalloc(&C, size_t(ld) * ld * 8);
    hipMalloc(&out, 64);
    {
        std::vector<double> h(size_t(K) * ld);
        unsigned s = 12345u;
        for (auto &x : h) {
            s = s * 1664525u + 1013904223u;
            x = (double(s >> 8) / double(1 << 24)) - 0.5;
        }
        hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice);
        for (auto &x : h) {
            s = s * 1664525u + 1013904223u;
            x = (double(s >> 8) / double(1 << 24)) - 0.5;
        }
        hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);

    printf("(0) back-to-back v_mfma_f64_16x16x4_f64, 8 accumulators per wave, ~40 ms launches\n");
    printf("| accumulators in | workgroups of 512 per CU | waves per SIMD | ms | TFLOP/s | cycles per MFMA and SIMD at 2.39 GHz |\n|---|---|---|---|---|---|\n");
    for (int agpr = 0; agpr < 2; ++agpr)
        for (int wgs : {1, 2}) {
            const int iters = 40000 / wgs;
            auto kern = agpr ? k_pure<1> : k_pure<0>;
            kern<<<cus * wgs, 512>>>(out, 10, 0.5);
            hipEventRecord(e0);
            kern<<<cus * wgs, 512>>>(out, iters, 0.5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = double(cus) * wgs * 8 * double(iters) * 8.0;   // MFMAs of the launch
            printf("| %s | %d | %d | %.2f | %.1f | %.1f |\n", agpr ? "AccVGPRs" : "VGPRs", wgs, 2 * wgs, ms, n * 2048.0 / (ms * 1e-3) / 1e12,
                   ms * 1e-3 * 2.39e9 / (n / (cus * 4.0)));
        }
    printf("\n");

    printf("(1) MFMA + LDS reads only (no global traffic, no barriers), ~20 ms launches, executed TFLOP/s\n");
    printf("| per wave FRM x FRN | threads | waves/SIMD budget | workgroups per CU launched | ms | TFLOP/s |\n|---|---|---|---|---|---|\n");
    auto lds_run = [&](const char *nm, auto kern, int nth, int wgs_per_cu, int frm, int frn, int minw) {
        const int iters = 6000;
        kern<<<cus * wgs_per_cu, nth>>>(out, 10);
        hipEventRecord(e0);
        kern<<<cus * wgs_per_cu, nth>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = double(cus) * wgs_per_cu * (nth / 64) * double(iters) * 4 * frm * frn * 2048.0;
        printf("| %s | %d | %d | %d | %.2f | %.1f |\n", nm, nth, minw, wgs_per_cu, ms, fl / (ms * 1e-3) / 1e12);
    };
    lds_run("2 x 4", k_lds_mfma<2, 4, 512, 4>, 512, 1, 2, 4, 4);
    lds_run("2 x 4", k_lds_mfma<2, 4, 512, 4>, 512, 2, 2, 4, 4);
    lds_run("4 x 4", k_lds_mfma<4, 4, 256, 2>, 256, 1, 4, 4, 2);
    lds_run("4 x 4", k_lds_mfma<4, 4, 256, 2>, 256, 2, 4, 4, 2);
    lds_run("4 x 4", k_lds_mfma<4, 4, 512, 2>, 512, 1, 4, 4, 2);

    printf("\n(2) whole GEMM, K = %d, blocks of 128 x 128 outputs: 256 | 512 | 595 | 1024 | 1296 (executed TFLOP/s of the launch)\n", K);
    printf("| variant | 16 x 16 | 16 x 32 | 17 x 35 | 32 x 32 | 36 x 36 |\n|---|---|---|---|---|---|\n");
    //           TMW TNW WR WC BK PIPE MINW
    run_variant<32, 64, 4, 2, 16, 0, 4>("product: 4 x 2 waves of 32 x 64, BK 16", A, B, C, ld, K, e0, e1);
    run_variant<32, 64, 4, 2, 16, 1, 4>("same, operand registers double-buffered", A, B, C, ld, K, e0, e1);
    run_variant<32, 64, 4, 2, 32, 0, 4>("same, BK 32", A, B, C, ld, K, e0, e1);
    run_variant<64, 64, 2, 2, 16, 0, 2>("2 x 2 waves of 64 x 64, BK 16", A, B, C, ld, K, e0, e1);
    run_variant<64, 64, 2, 2, 16, 1, 2>("2 x 2 waves of 64 x 64, BK 16, double-buffered", A, B, C, ld, K, e0, e1);
    run_variant<64, 64, 4, 2, 16, 0, 2>("256 x 128: 4 x 2 waves of 64 x 64, BK 16", A, B, C, ld, K, e0, e1);
    run_variant<64, 64, 4, 2, 16, 1, 2>("256 x 128: 4 x 2 waves of 64 x 64, BK 16, double-buffered", A, B, C, ld, K, e0, e1);
    run_variant<32, 64, 4, 2, 16, 0, 2>("product tile, register budget of 2 waves per SIMD", A, B, C, ld, K, e0, e1);
    return 0;
}
