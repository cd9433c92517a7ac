// f64 "TN" GEMM on v_mfma_f64_16x16x4_f64 for gfx950:
//     C[M,N] = alpha * sum_k A[k,m] * B[k,n] + beta * C
// Both operands are k-major (row = k), which is the natural layout of every Gram-type
// contraction on the pruning path: rows are samples, columns are channels/features
//   LASSO Gram   Q = Zc^T Zc            (K = S*n,  M = N = c)         lib/decompose.py:434 + Lasso.fit
//   refit Gram   G = Xc^T Xc, R = Xc^T Yc  (K = N samples)            lib/decompose.py:622 -> LinearRegression
//   Cholesky trailing updates / block solves (K = 128).
//
// Tiling: 128x128 output tile per 512-thread workgroup (8 waves as 4 x 2, 32x64 per wave = 2x4 MFMA
// tiles, 64 accumulator VGPRs: two workgroups per CU), BK = 16 per LDS stage, next stage prefetched into
// registers while the current one is multiplied; 64x64 tiles of 256 threads for the skinny K <= 512 products.  LDS rows are padded by 16 doubles so the
// two k-rows a 32-lane half reads land on disjoint bank halves (ds_read_b64, 64 banks).
// Small outputs are split along K (multiples of 8 splits; split z of every tile is placed
// on XCD z % 8 so the tiles that share a k-chunk share an L2); the partial sums are added in
// a fixed order -> bitwise run-to-run reproducible, no floating-point atomics.
//
// Tile counts and the chip (round 4, tools/ubench/gemm_probe.hip, profiles/r04_gemm_probe.md).  The main loop runs
// at 68-70 TFLOP/s (0.88-0.9 of the 77 the VGPR form of the instruction issues) when the tile count is a multiple of the
// 512 workgroup slots of the chip (2 per CU), and at 52 when it is 595 -- the refit Gram of a 472-channel layer: the 83
// tiles of the second round run alone on 83 CUs.  So the tiles beyond the last full round are split along K into
// s chunks of their own ("tail split": s = 2 .. 8, chosen so that the chunks fill one more round as evenly as possible),
// and the workgroup that finishes a tile's LAST chunk adds the s partial blocks in chunk order (arrival counter; the
// blocks cross the XCDs as sc1 stores / loads, no whole-L2 fences: see the epilogue) -- no second launch.  The same in-kernel reduction serves the uniform
// split when it has at most 8 chunks, and the lower-triangle products write the mirrored tile from the epilogue: the
// separate reduce and mirror launches of rounds 1-3 (each a dispatch a busy chip makes the chain wait for) are gone
// from every product with <= 8 chunks.
#include <algorithm>
#include <cstdlib>

#include "cp_common.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LPAD = 16;   // LDS rows of BM + LPAD = 144 doubles
constexpr int NTHREADS = 256;

__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

__host__ __device__ __forceinline__ void decode_tile(int tri, int tile, int tiles_n, int &ti, int &tj) {
    if (tri == CP_TRI_NONE) {
        ti = tile / tiles_n;
        tj = tile - ti * tiles_n;
        return;
    }
    // row-wise enumeration of the lower triangle: tile = a(a+1)/2 + b, b <= a
    int a = int((sqrtf(8.f * float(tile) + 1.f) - 1.f) * 0.5f);
    while ((a + 1) * (a + 2) / 2 <= tile) ++a;
    while (a * (a + 1) / 2 > tile) --a;
    int b = tile - a * (a + 1) / 2;
    if (tri == CP_TRI_LOWER_MIRROR) {
        ti = a;
        tj = b;
    } else {
        ti = b;
        tj = a;
    }
}

// Super-tile enumeration for launches with many tiles: the tile list is ordered super-block by super-block (SB x SB tiles,
// row-major inside a block; lower / upper triangle: the blocks on and below the diagonal, a diagonal block holding only its
// own triangle).  An XCD works on a contiguous range of this list and keeps 2 x 32 workgroups resident, i.e. about one
// 8 x 8 super-block: its tiles share 8 + 8 operand panels in the XCD's L2 instead of the 2-3 + tiles_n panels of a band of
// tile rows.  tiles_m = tile rows (TRI_NONE), tiles_n = tile columns (= tiles per side for the triangles).
constexpr int SB = 8;
__host__ __device__ __forceinline__ void decode_tile_blocked(int tri, int idx, int tiles_m, int tiles_n, int &ti, int &tj) {
    if (tri == CP_TRI_NONE) {
        const int per_row = SB * tiles_n;                      // tiles of a full block row
        int I = idx / per_row;
        const int nbr = (tiles_m + SB - 1) / SB;
        if (I > nbr - 1) I = nbr - 1;
        const int rem = idx - I * per_row;
        const int h = imin(SB, tiles_m - I * SB);
        const int J = rem / (h * SB);
        const int r2 = rem - J * h * SB;
        const int w = imin(SB, tiles_n - J * SB);
        ti = I * SB + r2 / w;
        tj = J * SB + r2 % w;
        return;
    }
    const int T = tiles_n;
    int I = 0, before = 0;                                     // block row I holds h S I + h (h + 1) / 2 tiles
    for (;; ++I) {
        const int h = imin(SB, T - I * SB);
        const int cnt = h * SB * I + h * (h + 1) / 2;
        if (idx < before + cnt) break;
        before += cnt;
    }
    const int h = imin(SB, T - I * SB);
    const int rem = idx - before;
    int a, b;
    if (rem < h * SB * I) {                                    // a full block left of the diagonal
        const int J = rem / (h * SB), r2 = rem - J * h * SB;
        a = I * SB + r2 / SB;
        b = J * SB + r2 % SB;
    } else {                                                   // the diagonal block: row r of it has r + 1 tiles
        const int r3 = rem - h * SB * I;
        int r = 0;
        while ((r + 1) * (r + 2) / 2 <= r3) ++r;
        a = I * SB + r;
        b = I * SB + r3 - r * (r + 1) / 2;
    }
    if (tri == CP_TRI_LOWER_MIRROR) {
        ti = a;
        tj = b;
    } else {
        ti = b;
        tj = a;
    }
}

// WT = per-wave output tile (64 -> 128x128 workgroup tile, 32 -> 64x64).  The small tile is
// used for the skinny K = 128 products of the blocked Cholesky / substitutions, where the
// large one would leave most CUs idle and make every call as long as one 128^3 tile.
// A second product of the same shape (other operands, other K, no split) can ride in the same launch
// as blockIdx.y == 1: one launch instead of two for the two LASSO Grams.
struct GemmSecond {
    const double *A, *B;
    double *C;
    int K;
};

// Gram and right-hand side of the normal equations in ONE launch (TRI == TRI_AUG, cp_gemm_gram_xty): the lower triangle of
// [X | Y]^T [X | Y] without its Y^T Y corner.  Tile rows ti < tp are rows of G = X^T X (lower tile + mirror, as
// CP_TRI_LOWER_MIRROR); tile rows ti >= tp take their A operand from Y (A2, lda2) and hold a tile of Y^T X = R^T, which is
// written transposed into R (C2, ldc2); the tiles with tj >= tp are not wanted and their workgroups leave at once.
struct GemmAug {
    const double *A2;
    int lda2, tp;
    double *C2;
    int ldc2;
};
constexpr int TRI_AUG = 3;

// How the workgroups of a launch divide the tiles (the tile list is ordered as decode_tile / decode_tile_blocked say):
//   units [0, n_full)                      tile `unit` computed whole, written straight to C
//   units [n_full, n_full + n_split * s)   tile n_full + t, k-chunk z: partial block -> Pb[t * s + z]; the last arrival
//                                          at cnt[t] adds the s blocks in chunk order and writes C
//   planes (legacy, > 8 chunks): every unit writes plane z of P (M x N each), k_gemm_reduce adds them
struct GemmSched {
    int n_full, n_split, s, kchunk;   // kchunk: k-rows per chunk of a split tile (a multiple of BK)
    double *Pb;                       // [n_split * s] blocks of TM x TM doubles
    int *cnt;                         // [n_split] arrival counters, zero before and after the launch
    int planes;                       // != 0: legacy plane mode with this many uniform splits
};

// unit (= workgroup id) of a launch -> its tile and k-chunk.  Host and device: cp_debug_gemm_units replays the kernel's own
// mapping for the CPU test that every tile / chunk of every shape is covered exactly once (tests/test_host_logic.py).
struct GemmUnit {
    int ti, tj, z, nz, t_split;
    bool idle;
};
__host__ __device__ __forceinline__ GemmUnit gemm_unit(int tri, int L, int n_tiles, int tiles_m, int tiles_n, const GemmSched &sch) {
    GemmUnit u{0, 0, 0, 1, -1, false};
    int tile;
    const bool blocked = !sch.planes && n_tiles >= 64;   // many tiles: the super-tile ordered list, an XCD per contiguous range
    if (sch.planes) {   // XCD-aware: consecutive workgroup ids round-robin over the 8 XCDs
        const int xcd = L & 7, slot = L >> 3;
        u.z = xcd + 8 * (slot / n_tiles);
        tile = slot % n_tiles;
        u.nz = sch.planes;
    } else if (L < sch.n_full) {
        if (blocked) {
            // workgroup L runs on XCD L % 8 (each XCD has its own L2).  Give every XCD a CONTIGUOUS range of the
            // super-tile ordered list (decode_tile_blocked): the workgroups resident on an XCD at any time then cover about one
            // 8 x 8 block of tiles and share its 16 operand panels in that L2 (PMC: the p = 4250 refit Gram fetched 2.8 GB for
            // 0.31 GB of operands with the plain order, 2.4 GB with bands of tile rows per XCD).
            const int xcd = L & 7, slot = L >> 3;
            const int base = sch.n_full >> 3, rem = sch.n_full & 7;
            tile = xcd * base + (xcd < rem ? xcd : rem) + slot;
        } else {
            tile = L;
        }
    } else {
        const int v = L - sch.n_full;
        u.nz = sch.s;
        if (sch.n_full == 0 && (sch.s & 7) == 0) {   // uniform split: chunk z of every tile on XCD z % 8
            const int xcd = L & 7, slot = L >> 3;
            u.z = xcd + 8 * (slot / sch.n_split);
            u.t_split = slot % sch.n_split;
        } else {
            // tail split: an XCD takes a contiguous range of the tail tiles (neighbours in the super-tile order) and runs
            // them chunk by chunk, so that the workgroups resident on it at any time share operand panels AND their k-range
            // in its L2 as the whole tiles did.  (Chunks of a tile next to each other in the grid -- each XCD a mix of tiles
            // and k-ranges -- made the tail round HBM-bound: the 666-tile Gram 2.39 ms against 2.17 without a tail split.)
            const int xcd = v & 7, slot = v >> 3;   // n_full is a multiple of the slot count: v & 7 is the XCD of the workgroup
            const int base = sch.n_split >> 3, rem = sch.n_split & 7;
            const int cnt = base + (xcd < rem ? 1 : 0);
            if (slot >= cnt * sch.s) {
                u.idle = true;
                return u;
            }
            u.z = slot / cnt;
            u.t_split = xcd * base + (xcd < rem ? xcd : rem) + (slot - u.z * cnt);
        }
        tile = sch.n_full + u.t_split;
    }
    if (blocked)
        decode_tile_blocked(tri, tile, tiles_m, tiles_n, u.ti, u.tj);
    else
        decode_tile(tri, tile, tiles_n, u.ti, u.tj);
    return u;
}

template <int TRI, int TAG, int WT, int NTH>
__global__ void __launch_bounds__(NTH, NTH == 512 ? 4 : 2)
k_gemm_tn_f64(int M, int N, int K, double alpha, const double *__restrict__ A, int lda,
              const double *__restrict__ B, int ldb, double beta, double *__restrict__ C, int ldc,
              double *__restrict__ P, int n_tiles, int tiles_n, GemmSched sch, GemmSecond second, GemmAug aug) {
    int kchunk = sch.kchunk;
    if (blockIdx.y == 1) {
        A = second.A;
        B = second.B;
        C = second.C;
        K = second.K;
        kchunk = second.K;
    }
    // NTH = 256: 2 x 2 waves of WT x WT each; NTH = 512: 4 x 2 waves of (WT/2) x WT each (same
    // workgroup tile, half the accumulators per wave -> 4 waves per SIMD and two workgroups per CU: while one waits at its
    // stage barrier the other one keeps the matrix pipe busy)
    constexpr int TM = 2 * WT;              // workgroup tile edge
    constexpr int TLD = TM + LPAD;          // padded LDS row
    constexpr int TPR = TM / 2;             // threads per tile row (one double2 each)
    constexpr int RPP = NTH / TPR;          // rows per pass
    constexpr int NPASS = BK / RPP;
    constexpr int WROWS = NTH / 128;        // wave rows (2 or 4)
    constexpr int WTM = TM / WROWS;         // wave tile rows
    constexpr int FRM = WTM / 16, FRN = WT / 16;  // MFMA tiles per wave (rows, cols)
    // two LDS stages: stage kt+1 is written while stage kt is being read, one barrier per k-stage
    __shared__ __attribute__((aligned(16))) double As2[2][BK * TLD];
    __shared__ __attribute__((aligned(16))) double Bs2[2][BK * TLD];
    __shared__ int s_last;

    const int tid = threadIdx.x;
    const int L = blockIdx.x;
    // ---- unit -> (tile, chunk z of nz) ----
    const GemmUnit un = gemm_unit(TRI == TRI_AUG ? CP_TRI_LOWER_MIRROR : TRI, L, n_tiles, M / TM, tiles_n, sch);
    if (un.idle) return;                             // the grid of a tail split is padded to the XCD with the most tiles
    const int ti = un.ti, tj = un.tj, z = un.z, nz = un.nz, t_split = un.t_split;
    const bool rhs_tile = TRI == TRI_AUG && ti >= aug.tp;      // a tile of Y^T X (workgroup-uniform)
    if (TRI == TRI_AUG && tj >= aug.tp) return;                // the Y^T Y corner: not wanted (its counters are never touched)
    if (rhs_tile) {
        A = aug.A2 - size_t(aug.tp) * TM;                      // column m0 of [X | Y] is column m0 - tp TM of Y
        lda = aug.lda2;
    }
    const int m0 = ti * TM, n0 = tj * TM;
    const int k0 = nz > 1 ? z * kchunk : 0;
    int k1 = nz > 1 ? k0 + kchunk : K;
    if (k1 > K) k1 = K;
    const int nk = k1 > k0 ? (k1 - k0) / BK : 0;

    // global -> register staging: thread loads rows (lrow + RPP i), 2 doubles at column lcol
    const int lrow = tid / TPR, lcol = (tid % TPR) * 2;
    const double *Ag = A + size_t(k0 + lrow) * lda + m0 + lcol;
    const double *Bg = B + size_t(k0 + lrow) * ldb + n0 + lcol;
    // native 2-vectors (HIP's double2 is a struct with unions: the staging arrays then stay in scratch memory and every
    // prefetched tile makes a round trip through it, with the global-load latency exposed in every k-stage)
    typedef double v2f64 __attribute__((ext_vector_type(2)));
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

    v4f64 acc[FRM][FRN];
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FRN; ++j) acc[i][j] = v4f64{0., 0., 0., 0.};

    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            *reinterpret_cast<v2f64 *>(&As2[buf][(lrow + RPP * i) * TLD + lcol]) = ar[i];
            *reinterpret_cast<v2f64 *>(&Bs2[buf][(lrow + RPP * i) * TLD + lcol]) = br[i];
        }
    };
    if (nk > 0) {
        load_tile(0);
        stage_write(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const double *As = As2[kt & 1], *Bs = Bs2[kt & 1];
        if (kt + 1 < nk) load_tile(kt + 1);
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
                for (int j = 0; j < FRN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stage_write((kt + 1) & 1);  // the other buffer: its readers passed the previous barrier
        __syncthreads();
    }

    // epilogue.  D layout of v_mfma_f64_16x16x4_f64: lane l, reg r -> row (l>>4) + 4r, col l&15.
    if (sch.planes) {
        double *dst = P + size_t(z) * M * N;
#pragma unroll
        for (int i = 0; i < FRM; ++i)
#pragma unroll
            for (int j = 0; j < FRN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm + i * 16 + fk + 4 * r, col = n0 + wn + j * 16 + fi;
                    dst[size_t(row) * N + col] = acc[i][j][r];
                }
        return;
    }
    if (nz > 1) {
        // partial block of chunk z, then the arrival counter of the tile: whoever arrives last adds the nz blocks in chunk
        // order (its own included: the sum does not depend on who that is) and goes on to the epilogue.
        // The blocks cross XCDs, i.e. L2s.  They are written and read with agent-scope relaxed atomics -- sc1 stores that
        // write through the L2, sc1 loads that do not hit it -- instead of plain accesses between an agent-scope release
        // and acquire: those fences are a write-back (per releasing wave) and an invalidation (per acquiring wave) of the
        // WHOLE L2 of the XCD, which cost this kernel 0.2-0.4 ms per launch and slowed every kernel running next to it
        // (vgg16 job 26.5 -> 30.5 ms, gpurun_out/r04_call23).  Each wave waits for its stores (explicit vmcnt(0), below) before the barrier, then
        // thread 0 counts the workgroup in.
        unsigned long long *blk = reinterpret_cast<unsigned long long *>(sch.Pb) + (size_t(t_split) * nz + z) * (TM * TM);
#pragma unroll
        for (int i = 0; i < FRM; ++i)
#pragma unroll
            for (int j = 0; j < FRN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(blk + (wm + i * 16 + fk + 4 * r) * TM + wn + j * 16 + fi,
                                       (unsigned long long)__double_as_longlong(acc[i][j][r]), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        // every wave waits for ITS OWN write-through stores to be acknowledged before the barrier: the compiler puts no
        // s_waitcnt vmcnt(0) between relaxed stores and s_barrier (checked in the gfx950 ISA: `llvm-objdump -d` of this
        // kernel shows `s_waitcnt vmcnt(0)` directly before the `s_barrier` below only with this statement), and without it
        // another wave's block could still be in flight when thread 0 counts the workgroup in.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CP_HANDOFF_RELEASE();              // nothing unless built with -DCP_HANDOFF_FENCES=1 (cp_common.h)
        __syncthreads();
        if (tid == 0) {
            const int prev = __hip_atomic_fetch_add(sch.cnt + t_split, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = prev == nz - 1;
            if (prev == nz - 1) __hip_atomic_store(sch.cnt + t_split, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
        CP_HANDOFF_ACQUIRE();
        // chunk by chunk, the 16 loads of a row of MFMA tiles in flight together (element by element the nz blocks would be
        // nz dependent memory latencies per element: measured 0.4 ms for eight blocks)
        const unsigned long long *b0 = reinterpret_cast<const unsigned long long *>(sch.Pb) + size_t(t_split) * nz * (TM * TM);
#pragma unroll
        for (int i = 0; i < FRM; ++i) {
            for (int zz = 0; zz < nz; ++zz) {
                const unsigned long long *bz = b0 + size_t(zz) * (TM * TM) + (wm + i * 16 + fk) * TM + wn + fi;
                double v[FRN][4];
#pragma unroll
                for (int j = 0; j < FRN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[j][r] = __longlong_as_double((long long)__hip_atomic_load(bz + 4 * r * TM + j * 16, __ATOMIC_RELAXED,
                                                                                     __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
                for (int j = 0; j < FRN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = zz ? acc[i][j][r] + v[j][r] : v[j][r];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < FRM; ++i)
#pragma unroll
        for (int j = 0; j < FRN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm + i * 16 + fk + 4 * r, col = n0 + wn + j * 16 + fi;
                double v = alpha * acc[i][j][r];
                if (TRI == TRI_AUG && rhs_tile) {      // (Y^T X)[row - tp TM, col] = R[col, row - tp TM]
                    aug.C2[size_t(col) * aug.ldc2 + (row - aug.tp * TM)] = v;
                    continue;
                }
                double *c = C + size_t(row) * ldc + col;
                if (beta != 0.0) v += beta * *c;
                *c = v;
                // lower-triangle product: the mirrored tile from the same registers (4 consecutive doubles per lane group)
                if ((TRI == CP_TRI_LOWER_MIRROR || TRI == TRI_AUG) && ti != tj) C[size_t(col) * ldc + row] = v;
            }
}

// Fixed-order reduction of the split-K partials: C = alpha * sum_z P[z] + beta * C.  One
// workgroup per 4-row strip of a 128x128 tile (32 strips per tile), one double2 per thread;
// TRI == lower-mirror also writes the transposed element.
constexpr int RSTRIPS = BM / 4;
template <int TRI>
__global__ void __launch_bounds__(NTHREADS)
k_gemm_reduce(int M, int N, double alpha, const double *__restrict__ P, int splits, double beta,
              double *__restrict__ C, int ldc, int tiles_n) {
    int ti, tj;
    decode_tile(TRI, blockIdx.x / RSTRIPS, tiles_n, ti, tj);
    const int strip = blockIdx.x % RSTRIPS;
    const int m0 = ti * BM, n0 = tj * BN;
    const size_t plane = size_t(M) * N;
    const int r = strip * 4 + threadIdx.x / (BN / 2), cc = (threadIdx.x % (BN / 2)) * 2;
    const size_t off = size_t(m0 + r) * N + n0 + cc;
    double2 s = {0., 0.};
    int z = 0;
    for (; z + 4 <= splits; z += 4) {  // 4 loads in flight, summed in index order
        const double2 v0 = *reinterpret_cast<const double2 *>(P + size_t(z) * plane + off);
        const double2 v1 = *reinterpret_cast<const double2 *>(P + size_t(z + 1) * plane + off);
        const double2 v2 = *reinterpret_cast<const double2 *>(P + size_t(z + 2) * plane + off);
        const double2 v3 = *reinterpret_cast<const double2 *>(P + size_t(z + 3) * plane + off);
        s.x = ((s.x + v0.x) + v1.x) + v2.x + v3.x;
        s.y = ((s.y + v0.y) + v1.y) + v2.y + v3.y;
    }
    for (; z < splits; ++z) {
        const double2 v = *reinterpret_cast<const double2 *>(P + size_t(z) * plane + off);
        s.x += v.x;
        s.y += v.y;
    }
    s.x *= alpha;
    s.y *= alpha;
    double *c = C + size_t(m0 + r) * ldc + n0 + cc;
    if (beta != 0.0) {
        s.x += beta * c[0];
        s.y += beta * c[1];
    }
    c[0] = s.x;
    c[1] = s.y;
    if (TRI == CP_TRI_LOWER_MIRROR && ti != tj) {
        C[size_t(n0 + cc) * ldc + m0 + r] = s.x;
        C[size_t(n0 + cc + 1) * ldc + m0 + r] = s.y;
    }
}

struct GemmPlan {
    int n_tiles, tiles_n;
    bool small;      // 64x64 tiles
    int planes;      // legacy plane mode: uniform splits (> 8), reduced by k_gemm_reduce; 0 otherwise
    int n_full, n_split, s, kchunk;   // see GemmSched
};

// chunks per tile for the r tiles left over after the full rounds of `slots` workgroups: the smallest s whose chunk rounds
// ceil(r s / slots) / s come within 10 % of the best of s = 1 .. smax; 1: leave the tiles whole
int tail_chunks(int r, int slots, int smax) {
    double best = 1.0;
    for (int s = 2; s <= smax; ++s) best = std::min(best, double((r * s + slots - 1) / slots) / s);
    if (best > 0.8) return 1;
    for (int s = 2; s <= smax; ++s)
        if (double((r * s + slots - 1) / slots) / s <= 1.1 * best) return s;
    return 1;
}

GemmPlan make_plan(const cp_ctx *ctx, int M, int N, int K, int tri, bool in_place = false, bool paired = false) {
    GemmPlan p;
    int tm = M / BM, tn = N / BN;
    const int big_tiles = tri == CP_TRI_NONE ? tm * tn : tm * (tm + 1) / 2;
    // skinny products (few large tiles, short K, no mirroring needed): quarter-size tiles
    // (never for in-place products C = A^T C: with more than one row tile per column strip, one
    //  workgroup would overwrite rows another one still has to read)
    p.small = !in_place && tri != CP_TRI_LOWER_MIRROR && K <= 512 && big_tiles <= ctx->cu_count;
    if (p.small) {
        tm *= 2;
        tn *= 2;
    }
    p.tiles_n = tn;
    p.n_tiles = tri == CP_TRI_NONE ? tm * tn : tm * (tm + 1) / 2;
    p.planes = 0;
    p.n_full = p.n_tiles;
    p.n_split = 0;
    p.s = 1;
    p.kchunk = K;
    const int nk = K / BK;
    // Workgroups to aim for when splitting K.  Every split writes a full partial plane (PMC: the
    // refit Gram wrote ~3x its input bytes with 16 planes), so stay near one workgroup per CU.
    const int target = ctx->cu_count * 6 / 4;
    const int slots = 2 * ctx->cu_count;   // resident 512-thread workgroups (72 KB of LDS, <= 128 VGPRs each)
    if (p.small || in_place || paired) return p;
    if (p.n_tiles < target / 2 && nk >= 16) {
        int splits = (target + p.n_tiles - 1) / p.n_tiles;
        splits = (splits + 7) / 8 * 8;
        const int max_splits = (nk / 4) / 8 * 8;  // keep >= 4 k-stages per split
        if (splits > max_splits) splits = max_splits;
        if (splits >= 8) {
            p.kchunk = (nk + splits - 1) / splits * BK;
            if (splits == 8) {   // the last arrival of a tile adds its eight blocks itself
                p.n_full = 0;
                p.n_split = p.n_tiles;
                p.s = 8;
            } else {
                p.planes = splits;
            }
        }
    } else if (p.n_tiles > slots && nk >= 32) {
        const int r = p.n_tiles % slots;
        const int s = r ? tail_chunks(r, slots, std::min(8, nk / 16)) : 1;   // >= 16 k-stages per chunk
        if (s > 1) {
            p.n_split = r;
            p.n_full = p.n_tiles - r;
            p.s = s;
            p.kchunk = (nk + s - 1) / s * BK;
        }
    }
    return p;
}

constexpr int CNT_RING = 16, CNT_PER_LAUNCH = 1024;   // arrival counters: a ring of regions, one per launch

}  // namespace

size_t cp_gemm_tn_workspace(const cp_ctx *ctx, int M, int N, int K, int tri) {
    GemmPlan p = make_plan(ctx, M, N, K, tri);
    if (p.planes) return size_t(p.planes) * M * N * sizeof(double) + 256;
    return p.n_split ? size_t(p.n_split) * p.s * BM * BN * sizeof(double) + 256 : 0;
}

static int gemm_launch(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                       int ldb, double beta, double *C, int ldc, int tri, const GemmSecond *second,
                       const GemmAug *aug = nullptr);

int cp_gemm_tn_f64(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                   int ldb, double beta, double *C, int ldc, int tri) {
    return gemm_launch(ctx, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, tri, nullptr);
}

// G = X^T X (p_pad x p_pad, both triangles) and R = X^T Y (p_pad x n_pad) in ONE launch: the lower triangle of the Gram of
// [X | Y] without its Y^T Y corner (GemmAug) -- R's tiles ride in the Gram's launch with the Gram's tile schedule instead of
// a second, skinny product (4 tile columns: a third of the chip, split along K and reduced) that a busy chip makes the
// layer wait for.  false in *fused: the shape's plan needs the legacy plane reduction (few tiles): the caller runs the
// two products one after the other as before.
int cp_gemm_gram_xty(cp_ctx *ctx, int p_pad, int n_pad, int K, const double *X, int ldx, const double *Y, int ldy, double *G,
                     int ldg, double *R, int ldr, bool *fused) {
    const int Ma = p_pad + n_pad;
    *fused = false;
    if (p_pad % BM || n_pad % BN || K % BK) return CP_OK;
    const GemmPlan p = make_plan(ctx, Ma, Ma, K, CP_TRI_LOWER_MIRROR);
    if (p.planes || p.small) return CP_OK;
    const GemmAug ag{Y, ldy, p_pad / BM, R, ldr};
    *fused = true;
    return gemm_launch(ctx, Ma, Ma, K, 1.0, X, ldx, X, ldx, 0.0, G, ldg, CP_TRI_LOWER_MIRROR, nullptr, &ag);
}
size_t cp_gemm_gram_xty_workspace(const cp_ctx *ctx, int p_pad, int n_pad, int K) {
    return cp_gemm_tn_workspace(ctx, p_pad + n_pad, p_pad + n_pad, K, CP_TRI_LOWER_MIRROR);
}

// C1 = alpha A1^T B1 (K1) and C2 = alpha A2^T B2 (K2), same M, N, leading dimensions and tri, in one launch when
// neither product needs split-K (else two launches)
int cp_gemm_tn_f64_pair(cp_ctx *ctx, int M, int N, double alpha, int K1, const double *A1, const double *B1, double *C1,
                        int K2, const double *A2, const double *B2, double *C2, int lda, int ldb, int ldc, int tri) {
    const GemmPlan p1 = make_plan(ctx, M, N, K1, tri), p2 = make_plan(ctx, M, N, K2, tri);
    if (!p1.planes && !p1.n_split && !p2.planes && !p2.n_split && p1.small == p2.small && K2 % BK == 0 && C1 != A1 &&
        C1 != B1) {
        const GemmSecond sec{A2, B2, C2, K2};
        return gemm_launch(ctx, M, N, K1, alpha, A1, lda, B1, ldb, 0.0, C1, ldc, tri, &sec);
    }
    CP_TRY(gemm_launch(ctx, M, N, K1, alpha, A1, lda, B1, ldb, 0.0, C1, ldc, tri, nullptr));
    return gemm_launch(ctx, M, N, K2, alpha, A2, lda, B2, ldb, 0.0, C2, ldc, tri, nullptr);
}

static int gemm_launch(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                       int ldb, double beta, double *C, int ldc, int tri, const GemmSecond *second, const GemmAug *aug) {
    if (M <= 0 || N <= 0) return CP_OK;
    if (M % BM || N % BN || K % BK || (lda & 1) || (ldb & 1) || (tri != CP_TRI_NONE && M != N))
        return cp_set_error(ctx, CP_ERR_ARG, "gemm_tn: unaligned shape M=%d N=%d K=%d lda=%d ldb=%d", M, N, K, lda,
                            ldb);
    const bool in_place = (C == A || C == B);
    if (in_place && (M > BM || tri != CP_TRI_NONE))
        return cp_set_error(ctx, CP_ERR_ARG, "gemm_tn: in-place product needs a single row tile (M <= %d)", BM);
    GemmPlan p = make_plan(ctx, M, N, K, tri, in_place, second != nullptr);
    double *P = nullptr;
    const size_t arena_mark = ctx->arena_used;  // P is transient: stream order makes reuse safe
    GemmSched sch{p.n_full, p.n_split, p.s, p.kchunk, nullptr, nullptr, p.planes};
    if (p.planes) {
        P = cp_arena_take_t<double>(ctx, size_t(p.planes) * M * N);
        if (!P) return cp_set_error(ctx, CP_ERR_NOMEM, "gemm_tn: arena exhausted (split-K partials)");
    } else if (p.n_split) {
        if (p.n_split > CNT_PER_LAUNCH) return cp_set_error(ctx, CP_ERR_ARG, "gemm_tn: %d split tiles", p.n_split);
        sch.Pb = cp_arena_take_t<double>(ctx, size_t(p.n_split) * p.s * BM * BN);
        if (!sch.Pb) return cp_set_error(ctx, CP_ERR_NOMEM, "gemm_tn: arena exhausted (split-K partial blocks)");
        if (!ctx->gemm_cnt) {   // zeroed once: every launch leaves its counters at zero again
            CP_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->gemm_cnt), size_t(CNT_RING) * CNT_PER_LAUNCH * sizeof(int)));
            CP_HIP(ctx, hipMemsetAsync(ctx->gemm_cnt, 0, size_t(CNT_RING) * CNT_PER_LAUNCH * sizeof(int), ctx->stream));
            CP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // once per context; whatever stream it is bound to later
        }
        sch.cnt = ctx->gemm_cnt + size_t(ctx->gemm_cnt_next) * CNT_PER_LAUNCH;
        ctx->gemm_cnt_next = (ctx->gemm_cnt_next + 1) % CNT_RING;
    }
    int units = p.planes ? p.n_tiles * p.planes : p.n_full + p.n_split * p.s;
    if (!p.planes && p.n_full > 0 && p.n_split > 0) units = p.n_full + 8 * ((p.n_split + 7) / 8) * p.s;   // tail: per-XCD ranges, padded
    const dim3 grid(units, second ? 2 : 1);
    const GemmSecond sec = second ? *second : GemmSecond{nullptr, nullptr, nullptr, 0};
    const GemmAug ag = aug ? *aug : GemmAug{nullptr, 0, 0, nullptr, 0};
#define CP_GEMM_LAUNCH(T, G)                                                                                  \
    do {                                                                                                      \
        if (p.small)                                                                                          \
            k_gemm_tn_f64<T, G, 32, 256><<<grid, 256, 0, ctx->stream>>>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                                                                          P, p.n_tiles, p.tiles_n, sch, sec, ag);   \
        else                                                                                                  \
            k_gemm_tn_f64<T, G, 64, 512><<<grid, 512, 0, ctx->stream>>>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                                                                          P, p.n_tiles, p.tiles_n, sch, sec, ag);   \
    } while (0)
    const int tag = ctx->gemm_tag;
    ctx->gemm_tag = CP_GEMM_GENERIC;
    if (tri == CP_TRI_NONE) {
        if (tag == CP_GEMM_REFIT_XTY)
            CP_GEMM_LAUNCH(CP_TRI_NONE, CP_GEMM_REFIT_XTY);
        else
            CP_GEMM_LAUNCH(CP_TRI_NONE, CP_GEMM_GENERIC);
    } else if (tri == CP_TRI_LOWER_MIRROR && aug) {
        k_gemm_tn_f64<TRI_AUG, CP_GEMM_REFIT_GRAM, 64, 512><<<grid, 512, 0, ctx->stream>>>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, P,
                                                                                        p.n_tiles, p.tiles_n, sch, sec, ag);
    } else if (tri == CP_TRI_LOWER_MIRROR) {
        if (tag == CP_GEMM_LASSO_GRAM)
            CP_GEMM_LAUNCH(CP_TRI_LOWER_MIRROR, CP_GEMM_LASSO_GRAM);
        else if (tag == CP_GEMM_REFIT_GRAM)
            CP_GEMM_LAUNCH(CP_TRI_LOWER_MIRROR, CP_GEMM_REFIT_GRAM);
        else
            CP_GEMM_LAUNCH(CP_TRI_LOWER_MIRROR, CP_GEMM_GENERIC);
    } else {
        CP_GEMM_LAUNCH(CP_TRI_UPPER, CP_GEMM_GENERIC);
    }
#undef CP_GEMM_LAUNCH
    CP_LAUNCH_CHECK(ctx);
    if (ctx->gemm_mark) {
        cp_stage_mark(ctx, ctx->gemm_mark);
        ctx->gemm_mark = nullptr;
    }
    if (p.planes) {
        if (tri == CP_TRI_NONE)
            k_gemm_reduce<CP_TRI_NONE><<<p.n_tiles * RSTRIPS, NTHREADS, 0, ctx->stream>>>(M, N, alpha, P, p.planes, beta, C,
                                                                                  ldc, p.tiles_n);
        else if (tri == CP_TRI_LOWER_MIRROR)
            k_gemm_reduce<CP_TRI_LOWER_MIRROR><<<p.n_tiles * RSTRIPS, NTHREADS, 0, ctx->stream>>>(M, N, alpha, P, p.planes,
                                                                                          beta, C, ldc, p.tiles_n);
        else
            k_gemm_reduce<CP_TRI_UPPER><<<p.n_tiles * RSTRIPS, NTHREADS, 0, ctx->stream>>>(M, N, alpha, P, p.planes, beta, C,
                                                                                   ldc, p.tiles_n);
        CP_LAUNCH_CHECK(ctx);
    }
    ctx->arena_used = arena_mark;
    return CP_OK;
}

// The plan of a launch and (units != null) the tile / chunk of every workgroup of it, computed on the HOST by the functions the
// kernel runs (make_plan, gemm_unit): lets a CPU test sweep every shape for "each tile exactly once, each split tile by
// exactly s chunks".  plan[9]: n_tiles, tiles_n, small, planes, n_full, n_split, s, kchunk, units.
// units[5 * u]: ti, tj, z, nz, idle.  Returns the number of units, or a negative error code.
extern "C" int cp_debug_gemm_units(int cu_count, int M, int N, int K, int tri, int32_t *plan, int32_t *units, int max_units) {
    if (cu_count <= 0 || M <= 0 || N <= 0 || K <= 0 || !plan) return -CP_ERR_ARG;
    if (M % BM || N % BN || K % BK || (tri != CP_TRI_NONE && M != N)) return -CP_ERR_ARG;
    cp_ctx tmp;
    tmp.cu_count = cu_count;
    const GemmPlan p = make_plan(&tmp, M, N, K, tri);
    int n_units = p.planes ? p.n_tiles * p.planes : p.n_full + p.n_split * p.s;
    if (!p.planes && p.n_full > 0 && p.n_split > 0) n_units = p.n_full + 8 * ((p.n_split + 7) / 8) * p.s;
    const int32_t pl[9] = {p.n_tiles, p.tiles_n, p.small ? 1 : 0, p.planes, p.n_full, p.n_split, p.s, p.kchunk, n_units};
    memcpy(plan, pl, sizeof(pl));
    if (units) {
        if (max_units < n_units) return -CP_ERR_ARG;
        const GemmSched sch{p.n_full, p.n_split, p.s, p.kchunk, nullptr, nullptr, p.planes};
        const int tm_edge = p.small ? 64 : 128;
        for (int L = 0; L < n_units; ++L) {
            const GemmUnit u = gemm_unit(tri, L, p.n_tiles, M / tm_edge, p.tiles_n, sch);
            int32_t *o = units + 5 * size_t(L);
            o[0] = u.ti;
            o[1] = u.tj;
            o[2] = u.z;
            o[3] = u.nz;
            o[4] = u.idle ? 1 : 0;
        }
    }
    return n_units;
}
