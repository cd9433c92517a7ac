// Can the f64 GEMM use the vector pipe NEXT TO the matrix pipe?  (profiles/r04_mfma_valu_mix.md: a SIMD that issues
// v_mfma_f64_16x16x4_f64 back to back at 64 cycles each still retires a v_fma_f64 every 6-8 cycles from other waves.)
//
// Hybrid wave: the 32 x 64 wave tile of csrc/gemm_f64.hip is 2 x 4 MFMA tiles of 16 x 16.  VT of the four tile columns go
// to the vector pipe instead: lane (fk, fi) holds a[i] = A[k = fk][m = 16 i + fi] and b[j] = B[k = fk][n = 16 j + fi] for
// the matrix instruction anyway, so with b[j] rotated by rho inside its row of 16 lanes (two v_mov_b32 with a DPP row
// rotation) the lane owns the k = fk part of C[16 i + fi][16 j + (fi + rho) % 16]:  16 v_fma_f64 per tile and k-group, no
// extra LDS traffic; the four k-residues (lane rows) are added once, in the epilogue.
//   part (1): the inner loop alone, operands from LDS, no global memory, no barriers
//   part (2): the whole GEMM (K = 4992) against the product's loop, with a difference check
// Build / run on the GPU box: hipcc --offload-arch=gfx950 -O3 gemm_hybrid.hip -o gemm_hybrid && ./gemm_hybrid
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int LPAD = 16, SB = 8, BK = 16;
constexpr int TM = 128, TLD = TM + LPAD, NTH = 512, FRM = 2, FRN = 4, WTM = 32, WT = 64;

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

template <int RHO>
__device__ __forceinline__ double rot16(double v) {   // the value of the lane RHO places away inside the row of 16 lanes
    if (RHO == 0) return v;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + (RHO & 15), 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + (RHO & 15), 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int RHO>
__device__ __forceinline__ int rot16i(int v) {
    if (RHO == 0) return v;
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + (RHO & 15), 0xf, 0xf, false);
}

// One rotation of the vector tile column: b rotated by RHO (two DPP moves into the fixed pair v[254:255]), then the FRM
// fused multiply-adds.  One asm statement, so that the scheduler cannot pull the 15 rotations of a k-group together (30 live
// registers, and the moves then run before the matrix instructions they should hide behind); volatile, like the matrix
// instruction below: the statements keep the order they are written in.
template <int RHO>
__device__ __forceinline__ void vrot_fma(const double (&a)[FRM], double bj, double (&vacc)[FRM][16]) {
    static_assert(FRM == 2, "two rows of MFMA tiles per wave");
    if constexpr (RHO == 0) {
        asm volatile("v_fma_f64 %0, %2, %4, %0\n\tv_fma_f64 %1, %3, %4, %1"
                     : "+v"(vacc[0][0]), "+v"(vacc[1][0])
                     : "v"(a[0]), "v"(a[1]), "v"(bj));
    } else {
        asm volatile("v_mov_b32_dpp v254, %4 row_ror:%6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b32_dpp v255, %5 row_ror:%6 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fma_f64 %0, %2, v[254:255], %0\n\t"
                     "v_fma_f64 %1, %3, v[254:255], %1"
                     : "+v"(vacc[0][RHO]), "+v"(vacc[1][RHO])
                     : "v"(a[0]), "v"(a[1]), "v"(__double2loint(bj)), "v"(__double2hiint(bj)), "n"(RHO)
                     : "v254", "v255");
    }
}
__device__ __forceinline__ void mfma_v(v4f64 &acc, double a, double b) {
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// a k-group of the hybrid wave: 6 matrix instructions (tile columns 0 .. 2), each followed by its share of the 16 rotations
// of tile column 3
__device__ __forceinline__ void kgroup_hybrid(const double (&a)[FRM], const double (&b)[FRN], v4f64 (&acc)[FRM][3],
                                              double (&vacc)[FRM][16]) {
    mfma_v(acc[0][0], a[0], b[0]);
    vrot_fma<0>(a, b[3], vacc); vrot_fma<1>(a, b[3], vacc); vrot_fma<2>(a, b[3], vacc);
    mfma_v(acc[0][1], a[0], b[1]);
    vrot_fma<3>(a, b[3], vacc); vrot_fma<4>(a, b[3], vacc); vrot_fma<5>(a, b[3], vacc);
    mfma_v(acc[0][2], a[0], b[2]);
    vrot_fma<6>(a, b[3], vacc); vrot_fma<7>(a, b[3], vacc); vrot_fma<8>(a, b[3], vacc);
    mfma_v(acc[1][0], a[1], b[0]);
    vrot_fma<9>(a, b[3], vacc); vrot_fma<10>(a, b[3], vacc); vrot_fma<11>(a, b[3], vacc);
    mfma_v(acc[1][1], a[1], b[1]);
    vrot_fma<12>(a, b[3], vacc); vrot_fma<13>(a, b[3], vacc);
    mfma_v(acc[1][2], a[1], b[2]);
    vrot_fma<14>(a, b[3], vacc); vrot_fma<15>(a, b[3], vacc);
}

// the vector tiles to memory: add the four k-residues, lane row fk stores the rotations rho % 4 == fk
template <int RHO>
__device__ __forceinline__ void vstore(const double (&vacc)[FRM][16], double *Crow0, int ldc, int fk, int fi) {
    const int src = rot16i<RHO>(fi);   // whose b this lane multiplied with at rotation RHO
#pragma unroll
    for (int i = 0; i < FRM; ++i) {
        double t = vacc[i][RHO];
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if ((RHO & 3) == fk) Crow0[size_t(16 * i + fi) * ldc + src] = t;
    }
    if constexpr (RHO + 1 < 16) vstore<RHO + 1>(vacc, Crow0, ldc, fk, fi);
}

// VT = number of tile columns (of FRN = 4) on the vector pipe: 0 = the product's loop
template <int VT, int MINW>
__global__ void __launch_bounds__(NTH, MINW)
k_gemm(int M, int N, int K, const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb,
       double *__restrict__ C, int ldc, int tiles_m, int tiles_n) {
    constexpr int TPR = TM / 2, RPP = NTH / TPR, NPASS = BK / RPP, FM = FRN - VT;
    __shared__ __attribute__((aligned(16))) double As2[2][BK * TLD];
    __shared__ __attribute__((aligned(16))) double Bs2[2][BK * TLD];
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
    const int m0 = ti * TM, n0 = tj * TM;
    const int nk = K / BK;
    const int lrow = tid / TPR, lcol = (tid % TPR) * 2;
    const double *Ag = A + size_t(lrow) * lda + m0 + lcol;
    const double *Bg = B + size_t(lrow) * ldb + n0 + lcol;
    v2f64 ar[NPASS], br[NPASS];
    auto load_tile = [&](int kt) {
        const double *a = Ag + size_t(kt) * BK * lda;
        const double *b = Bg + size_t(kt) * BK * ldb;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            ar[i] = *reinterpret_cast<const v2f64 *>(a + size_t(RPP * i) * lda);
            br[i] = *reinterpret_cast<const v2f64 *>(b + size_t(RPP * i) * ldb);
        }
    };
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WT;
    const int fk = lane >> 4, fi = lane & 15;
    v4f64 acc[FRM][VT == 1 ? 3 : (FM > 0 ? FM : 1)];
    double vacc[VT > 0 ? VT : 1][FRM][16];
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = v4f64{0., 0., 0., 0.};
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int i = 0; i < FRM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) vacc[v][i][r] = 0.0;
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            *reinterpret_cast<v2f64 *>(&As2[buf][(lrow + RPP * i) * TLD + lcol]) = ar[i];
            *reinterpret_cast<v2f64 *>(&Bs2[buf][(lrow + RPP * i) * TLD + lcol]) = br[i];
        }
    };
    load_tile(0);
    stage_write(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const double *As = As2[kt & 1], *Bs = Bs2[kt & 1];
        if (kt + 1 < nk) load_tile(kt + 1);
        if constexpr (VT == 1) {
            // the operands of k-group kk + 1 are requested before the (order-pinned) instructions of k-group kk
            double a[2][FRM], b[2][FRN];
#pragma unroll
            for (int i = 0; i < FRM; ++i) a[0][i] = As[fk * TLD + wm + i * 16 + fi];
#pragma unroll
            for (int j = 0; j < FRN; ++j) b[0][j] = Bs[fk * TLD + wn + j * 16 + fi];
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int i = 0; i < FRM; ++i) a[(kk + 1) & 1][i] = As[((kk + 1) * 4 + fk) * TLD + wm + i * 16 + fi];
#pragma unroll
                    for (int j = 0; j < FRN; ++j) b[(kk + 1) & 1][j] = Bs[((kk + 1) * 4 + fk) * TLD + wn + j * 16 + fi];
                }
                kgroup_hybrid(a[kk & 1], b[kk & 1], acc, vacc[0]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[FRM], b[FRN];
#pragma unroll
                for (int i = 0; i < FRM; ++i) a[i] = As[(kk * 4 + fk) * TLD + wm + i * 16 + fi];
#pragma unroll
                for (int j = 0; j < FRN; ++j) b[j] = Bs[(kk * 4 + fk) * TLD + wn + j * 16 + fi];
#pragma unroll
                for (int i = 0; i < FRM; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) stage_write((kt + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + fk + 4 * r, col = n0 + wn + j * 16 + fi;
                C[size_t(row) * ldc + col] = acc[i][j][r];
            }
#pragma unroll
    for (int v = 0; v < VT; ++v)
        vstore<0>(vacc[v], C + size_t(m0 + wm) * ldc + n0 + wn + 16 * (FM + v), ldc, fk, fi);
}

// the inner loop alone: operands from LDS, nothing else
template <int VT, int MINW>
__global__ void __launch_bounds__(NTH, MINW) k_inner(double *out, int iters) {
    constexpr int FM = FRN - VT;
    __shared__ double s[2 * BK * TLD];
    for (int i = threadIdx.x; i < 2 * BK * TLD; i += NTH) s[i] = 1e-3 * (i % 977);
    __syncthreads();
    const int lane = threadIdx.x & 63, fk = lane >> 4, fi = lane & 15, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * WTM, wn = (wave & 1) * WT;
    v4f64 acc[FRM][VT == 1 ? 3 : (FM > 0 ? FM : 1)];
    double vacc[VT > 0 ? VT : 1][FRM][16];
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = v4f64{0., 0., 0., 0.};
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int i = 0; i < FRM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) vacc[v][i][r] = 0.0;
    for (int it = 0; it < iters; ++it) {
        const double *As = s + (it & 1) * BK * TLD, *Bs = s + ((it & 1) ^ 1) * BK * TLD;
        if constexpr (VT == 1) {
            // the operands of k-group kk + 1 are requested before the (order-pinned) instructions of k-group kk
            double a[2][FRM], b[2][FRN];
#pragma unroll
            for (int i = 0; i < FRM; ++i) a[0][i] = As[fk * TLD + wm + i * 16 + fi];
#pragma unroll
            for (int j = 0; j < FRN; ++j) b[0][j] = Bs[fk * TLD + wn + j * 16 + fi];
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                if (kk + 1 < BK / 4) {
#pragma unroll
                    for (int i = 0; i < FRM; ++i) a[(kk + 1) & 1][i] = As[((kk + 1) * 4 + fk) * TLD + wm + i * 16 + fi];
#pragma unroll
                    for (int j = 0; j < FRN; ++j) b[(kk + 1) & 1][j] = Bs[((kk + 1) * 4 + fk) * TLD + wn + j * 16 + fi];
                }
                kgroup_hybrid(a[kk & 1], b[kk & 1], acc, vacc[0]);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double a[FRM], b[FRN];
#pragma unroll
                for (int i = 0; i < FRM; ++i) a[i] = As[(kk * 4 + fk) * TLD + wm + i * 16 + fi];
#pragma unroll
                for (int j = 0; j < FRN; ++j) b[j] = Bs[(kk * 4 + fk) * TLD + wn + j * 16 + fi];
#pragma unroll
                for (int i = 0; i < FRM; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    double sum = 0;
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
    for (int v = 0; v < VT; ++v)
#pragma unroll
        for (int i = 0; i < FRM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += vacc[v][i][r];
    if (sum == 12345.678) out[0] = sum;
}

__global__ void k_maxdiff(const double *a, const double *b, size_t n, double *out) {
    double m = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        m = fmax(m, fabs(a[i] - b[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(reinterpret_cast<unsigned long long *>(out), (unsigned long long)__double_as_longlong(m));
}

struct Shape {
    int tm, tn;
};

template <int VT, int MINW>
static void run_gemm(const char *name, const double *A, const double *B, double *C, const double *Cref, int ld, int K,
                     hipEvent_t e0, hipEvent_t e1, double *dout) {
    auto kern = k_gemm<VT, MINW>;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NTH, 0);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern));
    const Shape shapes[] = {{16, 16}, {16, 32}, {32, 32}, {36, 36}};
    printf("| %s (%d regs, %d wg/CU) |", name, fa.numRegs, occ);
    for (const Shape &sh : shapes) {
        const int M = sh.tm * 128, N = sh.tn * 128;
        std::vector<float> t;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            kern<<<sh.tm * sh.tn, NTH>>>(M, N, K, A, ld, B, ld, C, ld, sh.tm, sh.tn);
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
    if (Cref) {   // the last shape filled the whole 4608 x 4608 C
        hipMemset(dout, 0, 8);
        k_maxdiff<<<1024, 256>>>(C, Cref, size_t(ld) * ld, dout);
        double d;
        hipMemcpy(&d, dout, 8, hipMemcpyDeviceToHost);
        printf(" max |C - C_product| = %.3e |", d);
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
    double *A, *B, *C, *Cref, *out;
    hipMalloc(&A, size_t(K) * ld * 8);
    hipMalloc(&B, size_t(K) * ld * 8);
    hipMalloc(&C, size_t(ld) * ld * 8);
    hipMalloc(&Cref, size_t(ld) * ld * 8);
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

    printf("(1) inner loop alone (LDS operands only), 512-thread workgroups, ~20 ms launches; TFLOP/s counts all 8 tiles of a wave\n");
    printf("| tile columns on the vector pipe | register budget (waves/SIMD) | workgroups per CU | ms | TFLOP/s | cycles per k-group and SIMD at 2.38 GHz |\n|---|---|---|---|---|---|\n");
    auto inner = [&](const char *nm, auto kern, int wgs, int minw) {
        const int iters = 6000 / wgs;
        kern<<<cus * wgs, NTH>>>(out, 10);
        hipEventRecord(e0);
        kern<<<cus * wgs, NTH>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double kgroups = double(cus) * wgs * 8 * double(iters) * 4;   // per wave: 4 k-groups per iteration
        printf("| %s | %d | %d | %.2f | %.1f | %.0f |\n", nm, minw, wgs, ms, kgroups * 8 * 2048.0 / (ms * 1e-3) / 1e12,
               ms * 1e-3 * 2.38e9 / (kgroups / (cus * 4.0)));
    };
    inner("0 (8 MFMA)", k_inner<0, 4>, 2, 4);
    inner("0 (8 MFMA)", k_inner<0, 2>, 1, 2);
    inner("1 (6 MFMA + 32 FMA + 30 DPP moves)", k_inner<1, 2>, 1, 2);

    printf("\n(2) whole GEMM, K = %d, blocks of 128 x 128 outputs: 256 | 512 | 1024 | 1296 (executed TFLOP/s of the launch)\n", K);
    printf("| variant | 16 x 16 | 16 x 32 | 32 x 32 | 36 x 36 | check |\n|---|---|---|---|---|---|\n");
    run_gemm<0, 4>("product loop", A, B, Cref, nullptr, ld, K, e0, e1, out);
    run_gemm<0, 2>("product loop, 2 waves per SIMD budget", A, B, C, Cref, ld, K, e0, e1, out);
    run_gemm<1, 2>("1 tile column of 4 on the vector pipe", A, B, C, Cref, ld, K, e0, e1, out);
    return 0;
}
