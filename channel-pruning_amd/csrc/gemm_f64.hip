// f64 "TN" GEMM on v_mfma_f64_16x16x4_f64 for gfx950:
//     C[M,N] = alpha * sum_k A[k,m] * B[k,n] + beta * C
// Both operands are k-major (row = k), which is the natural layout of every Gram-type
// contraction on the pruning path: rows are samples, columns are channels/features
//   LASSO Gram   Q = Zc^T Zc            (K = S*n,  M = N = c)         lib/decompose.py:434 + Lasso.fit
//   refit Gram   G = Xc^T Xc, R = Xc^T Yc  (K = N samples)            lib/decompose.py:622 -> LinearRegression
//   Cholesky trailing updates / block solves (K = 128).
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves, 64x64 per wave = 4x4 MFMA
// tiles, 128 accumulator VGPRs), BK = 16 per LDS stage, next stage prefetched into
// registers while the current one is multiplied.  LDS rows are padded by 16 doubles so the
// two k-rows a 32-lane half reads land on disjoint bank halves (ds_read_b64, 64 banks).
// Small outputs are split along K (multiples of 8 splits; split z of every tile is placed
// on XCD z % 8 so the tiles that share a k-chunk share an L2) and reduced by a second
// kernel in a fixed order -> bitwise run-to-run reproducible, no atomics.
#include <cstdlib>

#include "cp_common.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LPAD = 16, LLD = BM + LPAD;  // LLD = 144
constexpr int NTHREADS = 256;

__device__ __forceinline__ void decode_tile(int tri, int tile, int tiles_n, int &ti, int &tj) {
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
__device__ __forceinline__ void decode_tile_blocked(int tri, int idx, int tiles_m, int tiles_n, int &ti, int &tj) {
    if (tri == CP_TRI_NONE) {
        const int per_row = SB * tiles_n;                      // tiles of a full block row
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
        return;
    }
    const int T = tiles_n;
    int I = 0, before = 0;                                     // block row I holds h S I + h (h + 1) / 2 tiles
    for (;; ++I) {
        const int h = min(SB, T - I * SB);
        const int cnt = h * SB * I + h * (h + 1) / 2;
        if (idx < before + cnt) break;
        before += cnt;
    }
    const int h = min(SB, T - I * SB);
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

template <int TRI, int TAG, int WT, int NTH>
__global__ void __launch_bounds__(NTH, NTH == 512 ? 4 : 2)
k_gemm_tn_f64(int M, int N, int K, double alpha, const double *__restrict__ A, int lda,
              const double *__restrict__ B, int ldb, double beta, double *__restrict__ C, int ldc,
              double *__restrict__ P, int splits, int kchunk, int n_tiles, int tiles_n, GemmSecond second) {
    if (blockIdx.y == 1) {
        A = second.A;
        B = second.B;
        C = second.C;
        K = second.K;
        kchunk = second.K;
    }
    // NTH = 256: 2 x 2 waves of WT x WT each; NTH = 512: 4 x 2 waves of (WT/2) x WT each (same
    // workgroup tile, half the accumulators per wave -> 4 waves per SIMD, which is what it takes to
    // keep the f64 MFMA pipe busy: one wave alone issues one v_mfma_f64_16x16x4 per ~140 cycles)
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

    const int tid = threadIdx.x;
    // Persistent form: a launch may carry fewer workgroups than tiles (gridDim.x a multiple of 8, so that a workgroup's
    // virtual ids L keep its XCD); every workgroup then walks the tile list with stride gridDim.x.  A launch whose
    // workgroups are all resident leaves nothing queued in the dispatcher behind which other streams' kernels would wait.
    const int total_blocks = n_tiles * splits;
    for (int L = blockIdx.x; L < total_blocks; L += gridDim.x) {
    int tile, z;
    if (splits > 1) {  // XCD-aware: consecutive workgroup ids round-robin over the 8 XCDs
        const int xcd = L & 7, slot = L >> 3;
        z = xcd + 8 * (slot / n_tiles);
        tile = slot % n_tiles;
    } else if (n_tiles >= 64) {
        // XCD-aware: workgroup L runs on XCD L % 8 (each XCD has its own L2).  Give every XCD a CONTIGUOUS range of the
        // super-tile ordered list (decode_tile_blocked): the workgroups resident on an XCD at any time then cover about one
        // 8 x 8 block of tiles and share its 16 operand panels in that L2 (PMC: the p = 4250 refit Gram fetched 2.8 GB for
        // 0.31 GB of operands with the plain order, 2.4 GB with bands of tile rows per XCD).
        const int xcd = L & 7, slot = L >> 3;
        const int base = n_tiles >> 3, rem = n_tiles & 7;
        tile = xcd * base + (xcd < rem ? xcd : rem) + slot;
        z = 0;
    } else {
        tile = L;
        z = 0;
    }
    int ti, tj;
    if (splits == 1 && n_tiles >= 64)
        decode_tile_blocked(TRI, tile, M / TM, tiles_n, ti, tj);
    else
        decode_tile(TRI, tile, tiles_n, ti, tj);
    const int m0 = ti * TM, n0 = tj * TM;
    const int k0 = z * kchunk;
    int k1 = k0 + kchunk;
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
    if (splits > 1) {
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
    } else {
#pragma unroll
        for (int i = 0; i < FRM; ++i)
#pragma unroll
            for (int j = 0; j < FRN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm + i * 16 + fk + 4 * r, col = n0 + wn + j * 16 + fi;
                    double v = alpha * acc[i][j][r];
                    double *c = C + size_t(row) * ldc + col;
                    if (beta != 0.0) v += beta * *c;
                    *c = v;
                }
    }
    }   // persistent tile loop
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

// splits == 1 lower-triangle product: copy tile (ti,tj), tj < ti, to (tj,ti) transposed.
__global__ void __launch_bounds__(NTHREADS) k_mirror_lower(double *__restrict__ C, int ldc, int tiles_n) {
    __shared__ double t[32][33];
    int ti, tj;
    decode_tile(CP_TRI_LOWER_MIRROR, blockIdx.x, tiles_n, ti, tj);
    if (ti == tj) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int br = 0; br < BM; br += 32)
        for (int bc = 0; bc < BN; bc += 32) {
            for (int y = ty; y < 32; y += 8) t[y][tx] = C[size_t(ti * BM + br + y) * ldc + tj * BN + bc + tx];
            __syncthreads();
            for (int y = ty; y < 32; y += 8) C[size_t(tj * BN + bc + y) * ldc + ti * BM + br + tx] = t[tx][y];
            __syncthreads();
        }
}

struct GemmPlan {
    int n_tiles, tiles_n, splits, kchunk;
    bool small;  // 64x64 tiles
};

GemmPlan make_plan(const cp_ctx *ctx, int M, int N, int K, int tri, bool in_place = false) {
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
    const int nk = K / BK;
    // Workgroups to aim for when splitting K.  Every split writes a full partial plane (PMC: the
    // refit Gram wrote ~3x its input bytes with 16 planes), so stay near one workgroup per CU.
    const int target = ctx->cu_count * 6 / 4;
    int splits = 1;
    if (!p.small && p.n_tiles < target / 2 && nk >= 16) {
        splits = (target + p.n_tiles - 1) / p.n_tiles;
        splits = (splits + 7) / 8 * 8;
        const int max_splits = (nk / 4) / 8 * 8;  // keep >= 4 k-stages per split
        if (splits > max_splits) splits = max_splits;
        if (splits < 8) splits = 1;
    }
    if (splits > 1) {
        const int per = (nk + splits - 1) / splits;
        p.kchunk = per * BK;
    } else {
        p.kchunk = K;
    }
    p.splits = splits;
    return p;
}

}  // namespace

size_t cp_gemm_tn_workspace(const cp_ctx *ctx, int M, int N, int K, int tri) {
    GemmPlan p = make_plan(ctx, M, N, K, tri);
    return p.splits > 1 ? size_t(p.splits) * M * N * sizeof(double) + 256 : 0;
}

static int gemm_launch(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                       int ldb, double beta, double *C, int ldc, int tri, const GemmSecond *second);

int cp_gemm_tn_f64(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                   int ldb, double beta, double *C, int ldc, int tri) {
    return gemm_launch(ctx, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, tri, nullptr);
}

// C1 = alpha A1^T B1 (K1) and C2 = alpha A2^T B2 (K2), same M, N, leading dimensions and tri, in one launch when
// neither product needs split-K (else two launches)
int cp_gemm_tn_f64_pair(cp_ctx *ctx, int M, int N, double alpha, int K1, const double *A1, const double *B1, double *C1,
                        int K2, const double *A2, const double *B2, double *C2, int lda, int ldb, int ldc, int tri) {
    const GemmPlan p1 = make_plan(ctx, M, N, K1, tri), p2 = make_plan(ctx, M, N, K2, tri);
    if (p1.splits == 1 && p2.splits == 1 && p1.small == p2.small && K2 % BK == 0 && C1 != A1 && C1 != B1) {
        const GemmSecond sec{A2, B2, C2, K2};
        return gemm_launch(ctx, M, N, K1, alpha, A1, lda, B1, ldb, 0.0, C1, ldc, tri, &sec);
    }
    CP_TRY(gemm_launch(ctx, M, N, K1, alpha, A1, lda, B1, ldb, 0.0, C1, ldc, tri, nullptr));
    return gemm_launch(ctx, M, N, K2, alpha, A2, lda, B2, ldb, 0.0, C2, ldc, tri, nullptr);
}

static int gemm_launch(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda, const double *B,
                       int ldb, double beta, double *C, int ldc, int tri, const GemmSecond *second) {
    if (M <= 0 || N <= 0) return CP_OK;
    if (M % BM || N % BN || K % BK || (lda & 1) || (ldb & 1) || (tri != CP_TRI_NONE && M != N))
        return cp_set_error(ctx, CP_ERR_ARG, "gemm_tn: unaligned shape M=%d N=%d K=%d lda=%d ldb=%d", M, N, K, lda,
                            ldb);
    const bool in_place = (C == A || C == B);
    if (in_place && (M > BM || tri != CP_TRI_NONE))
        return cp_set_error(ctx, CP_ERR_ARG, "gemm_tn: in-place product needs a single row tile (M <= %d)", BM);
    GemmPlan p = make_plan(ctx, M, N, K, tri, in_place);
    double *P = nullptr;
    const size_t arena_mark = ctx->arena_used;  // P is transient: stream order makes reuse safe
    if (p.splits > 1) {
        P = cp_arena_take_t<double>(ctx, size_t(p.splits) * M * N);
        if (!P) return cp_set_error(ctx, CP_ERR_NOMEM, "gemm_tn: arena exhausted (split-K partials)");
    }
    // (The kernel walks the tile list with stride gridDim.x, so a launch may carry fewer workgroups than tiles.  Capping
    //  the grid at 64 .. 512 resident workgroups -- nothing left queued in the dispatcher behind which other streams'
    //  kernels would wait -- was measured in round 4: vgg16 job 46.4 / 35.4 / 31.9 / 31.0 ms at 64 / 128 / 256 / 512
    //  against 30.7 with one workgroup per tile; not used.)
    const dim3 grid(p.n_tiles * p.splits, second ? 2 : 1);
    const GemmSecond sec = second ? *second : GemmSecond{nullptr, nullptr, nullptr, 0};
#define CP_GEMM_LAUNCH(T, G)                                                                                  \
    do {                                                                                                      \
        if (p.small)                                                                                          \
            k_gemm_tn_f64<T, G, 32, 256><<<grid, 256, 0, ctx->stream>>>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                                                                          P, p.splits, p.kchunk, p.n_tiles, p.tiles_n, sec); \
        else                                                                                                  \
            k_gemm_tn_f64<T, G, 64, 512><<<grid, 512, 0, ctx->stream>>>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, \
                                                                          P, p.splits, p.kchunk, p.n_tiles, p.tiles_n, sec); \
    } while (0)
    const int tag = ctx->gemm_tag;
    ctx->gemm_tag = CP_GEMM_GENERIC;
    if (tri == CP_TRI_NONE) {
        if (tag == CP_GEMM_REFIT_XTY)
            CP_GEMM_LAUNCH(CP_TRI_NONE, CP_GEMM_REFIT_XTY);
        else
            CP_GEMM_LAUNCH(CP_TRI_NONE, CP_GEMM_GENERIC);
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
    if (p.splits > 1) {
        if (tri == CP_TRI_NONE)
            k_gemm_reduce<CP_TRI_NONE><<<p.n_tiles * RSTRIPS, NTHREADS, 0, ctx->stream>>>(M, N, alpha, P, p.splits, beta, C,
                                                                                  ldc, p.tiles_n);
        else if (tri == CP_TRI_LOWER_MIRROR)
            k_gemm_reduce<CP_TRI_LOWER_MIRROR><<<p.n_tiles * RSTRIPS, NTHREADS, 0, ctx->stream>>>(M, N, alpha, P, p.splits,
                                                                                          beta, C, ldc, p.tiles_n);
        else
            k_gemm_reduce<CP_TRI_UPPER><<<p.n_tiles * RSTRIPS, NTHREADS, 0, ctx->stream>>>(M, N, alpha, P, p.splits, beta, C,
                                                                                   ldc, p.tiles_n);
        CP_LAUNCH_CHECK(ctx);
    } else if (tri == CP_TRI_LOWER_MIRROR && p.n_tiles > 1) {
        k_mirror_lower<<<p.n_tiles, NTHREADS, 0, ctx->stream>>>(C, ldc, p.tiles_n);
        CP_LAUNCH_CHECK(ctx);
    }
    ctx->arena_used = arena_mark;
    return CP_OK;
}
