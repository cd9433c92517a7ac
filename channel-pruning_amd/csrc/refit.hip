// Least-squares reconstruction on the kept channels, replacing
//   fc_kernel(X[:, idxs].reshape(N, -1), Y)                      lib/decompose.py:622, 636-669
//   -> LinearRegression(fit_intercept=True).fit -> scipy.linalg.lstsq (gelsd)
//                                                                 sklearn/linear_model/_base.py:591-706
// i.e. centre X and Y by their column means, coef = argmin ||Xc coef^T - Yc||_F (minimum norm),
// intercept = ybar - xbar . coef^T (_base.py:300-316); ridge > 0 = the Ridge branch (decompose.py:662).
//
// Device formulation (p = kept * kk columns, all float64):
//   column means (two-stage deterministic reduction)                         HBM bound
//   Xs = gather(X, kept channels) - xbar,  Yc = Y - ybar  (tile-padded)      HBM bound
//   G = Xs^T Xs (p x p, f64 MFMA, symmetric half),  R = Xs^T Yc (p x n)      MFMA bound
//   Cholesky G = U^T U, blocked by 128, one launch per step with the right-hand sides riding along
//   (chol_step.hip: lazy trailing update, forward substitution for free)     MFMA / chain bound
//   block backward substitution with the inverted diagonal blocks            MFMA / launch bound
// The normal equations square the condition number, so they are only trusted while every Cholesky pivot stays
// above PIV_TOL (1e-6) of its original diagonal (error about eps / pivot ratio <= 1e-9).  Otherwise, and when
// N - 1 < p, the solve is redone by refit_robust() below, which is as accurate as the reference's gelsd
// (error ~ cond * eps, not cond^2 * eps) and reproduces its rank decision sigma_i <= max(N,p) * eps * sigma_max
// (sklearn/linear_model/_base.py:700-702 -> scipy.linalg.lstsq, decompose.py:665-666):
//   stage 1  R1 = chol(G + s I), s = 4 max(N,p) eps ||G||_inf; X1 = Xs R1^-1 explicitly (forward substitution on
//            Xs^T with N right-hand sides): a shifted-Cholesky-QR step, cond(X1) ~ sqrt(cond(Xs)^2 s) at worst
//   stage 2  G2 = X1^T X1, C2 = X1^T Yc from the preconditioned rows
//     path 1 (full column rank: chol(G2) keeps every pivot above 1e-11): Z = G2^-1 C2, W = R1^-1 Z
//     path 2 (numerically rank deficient: copies of channels, N <= p, ...): G2 = V T^2 V^T (one-sided Jacobi),
//            Xs = (X1 V T^-1)(T V^T R1) = Q B with Q orthonormal, B = Ub S Vb^T (one-sided Jacobi on the rows of B):
//            the singular values S ARE those of Xs; W = Vb S^+ Ub^T Q^T Yc over S_i > max(N,p) eps S_0 --
//            the minimum-norm solution and the rank gelsd reports.
// The ridge branch and the row-sharded tail (no access to the rows) keep the older fallback: iterated Tikhonov
//   W_{k+1} = W_k + (G + eI)^-1 (R - G W_k),  e = 1e-9 max diag(G),  5 sweeps (exact for spectra with a clean gap).
#include "cp_common.h"

namespace {

constexpr int RT = 256;
constexpr int NB = 128;  // Cholesky block = GEMM tile

template <typename T>
__device__ __forceinline__ double ldv(const T *p, size_t i) {
    return double(p[i]);
}

// ---- column means, gather + centre -------------------------------------------------------
// Column sums of the kept X columns (gathered through chan[]) and of Y in one launch:
// blockIdx.x < gx covers 256 X columns, the rest 256 Y columns; blockIdx.y = row block.
constexpr int CSU = 8;
template <typename TX>
__global__ void __launch_bounds__(RT) k_colsum_xy(const TX *__restrict__ X, const double *__restrict__ Y, int64_t N,
                                                  int c, int kk, int n, const int *__restrict__ chan, int p, int gx,
                                                  int rows_per_block, double *__restrict__ part_x, int ldx,
                                                  double *__restrict__ part_y, int ldy) {
    const int64_t r0 = int64_t(blockIdx.y) * rows_per_block, r1 = min(N, r0 + rows_per_block);
    double s = 0;
    if (int(blockIdx.x) < gx) {
        const int col = blockIdx.x * RT + threadIdx.x;
        if (col >= p) return;
        const int a = col / kk, t = col - a * kk;
        const size_t src = size_t(chan[a]) * kk + t, stride = size_t(c) * kk;
        // eight independent row loads in flight per thread, summed in row order (one dependent load per iteration left the
        // pass at 6 % of the HBM rate)
        int64_t r = r0;
        for (; r + CSU <= r1; r += CSU) {
            double v[CSU];
#pragma unroll
            for (int u = 0; u < CSU; ++u) v[u] = ldv(X, size_t(r + u) * stride + src);
#pragma unroll
            for (int u = 0; u < CSU; ++u) s += v[u];
        }
        for (; r < r1; ++r) s += ldv(X, size_t(r) * stride + src);
        part_x[size_t(blockIdx.y) * ldx + col] = s;
    } else {
        const int col = (blockIdx.x - gx) * RT + threadIdx.x;
        if (col >= n) return;
        int64_t r = r0;
        for (; r + CSU <= r1; r += CSU) {
            double v[CSU];
#pragma unroll
            for (int u = 0; u < CSU; ++u) v[u] = Y[size_t(r + u) * n + col];
#pragma unroll
            for (int u = 0; u < CSU; ++u) s += v[u];
        }
        for (; r < r1; ++r) s += Y[size_t(r) * n + col];
        part_y[size_t(blockIdx.y) * ldy + col] = s;
    }
}

// xmean / ymean from the row-block partials (fixed order)
__global__ void __launch_bounds__(RT) k_mean_finish_xy(const double *__restrict__ part_x, int ldx, int p,
                                                       const double *__restrict__ part_y, int ldy, int n, int gx,
                                                       int nparts, double inv_n, double *__restrict__ xmean,
                                                       double *__restrict__ ymean) {
    const bool is_x = int(blockIdx.x) < gx;
    const int col = (is_x ? blockIdx.x : blockIdx.x - gx) * RT + threadIdx.x;
    if (col >= (is_x ? p : n)) return;
    const double *part = is_x ? part_x : part_y;
    const int ldp = is_x ? ldx : ldy;
    double s = 0;
    for (int b = 0; b < nparts; ++b) s += part[size_t(b) * ldp + col];
    (is_x ? xmean : ymean)[col] = s * inv_n;
}

// Xs[r, :] = X[r, kept columns] - xmean and Yc[r, :] = Y[r, :] - ymean, zero padded (row r = blockIdx.x)
template <typename TX>
__global__ void __launch_bounds__(RT) k_gather_center_xy(const TX *__restrict__ X, const double *__restrict__ Y,
                                                         int64_t N, int c, int kk, int n, const int *__restrict__ chan,
                                                         int p, int p_pad, int n_pad, const double *__restrict__ xmean,
                                                         const double *__restrict__ ymean, double *__restrict__ Xs,
                                                         double *__restrict__ Yc) {
    const int64_t r = blockIdx.x;
    const size_t stride = size_t(c) * kk;
    for (int col = threadIdx.x; col < p_pad; col += RT) {
        double v = 0.0;
        if (r < N && col < p) {
            const int a = col / kk, t = col - a * kk;
            v = ldv(X, size_t(r) * stride + size_t(chan[a]) * kk + t) - xmean[col];
        }
        Xs[size_t(r) * p_pad + col] = v;
    }
    for (int col = threadIdx.x; col < n_pad; col += RT)
        Yc[size_t(r) * n_pad + col] = (r < N && col < n) ? Y[size_t(r) * n + col] - ymean[col] : 0.0;
}

// The kept-channel list as a bit mask in the kernel arguments (c <= 2048 -> 256 B of kernarg), unpacked by the kernel that
// needs the list: no upload, no launch that builds it (k_chan_from_bits for the kernels that read a list from memory,
// k_gather_normal_eq<true> into LDS).
constexpr int CHAN_BITS_MAX = 2048;
struct ChanBits {
    unsigned long long w[CHAN_BITS_MAX / 64];
};
__device__ __forceinline__ void chan_from_bits(const ChanBits &bits, int c, int *chan) {   // 256 threads x 8 channels
    const int c0 = threadIdx.x * 8;          // inside one 64-bit word
    if (c0 >= c) return;
    const int wi = c0 >> 6, sh = c0 & 63;
    int pos = 0;
    for (int i = 0; i < wi; ++i) pos += __popcll(bits.w[i]);
    pos += __popcll(bits.w[wi] & ((1ull << sh) - 1ull));
    const unsigned m = unsigned(bits.w[wi] >> sh) & 0xffu;
    for (int b = 0; b < 8 && c0 + b < c; ++b)
        if (m & (1u << b)) chan[pos++] = c0 + b;
}

// ---- diagonal handling -------------------------------------------------------------------
// dg0[i] = G[i,i] (original), gmax[0] = max_i dg0[i]; pad rows get G[i,i] = 1.
__global__ void __launch_bounds__(256) k_diag_prepare(double *__restrict__ G, int ld, int p, int p_pad, double ridge,
                                                       double *__restrict__ dg0, double *__restrict__ gmax,
                                                       int *__restrict__ info, int n_info) {
    __shared__ double red[16];
    for (int i = threadIdx.x; i < n_info; i += blockDim.x) info[i] = 0;  // [0] first failed pivot, [1 + b] block b done
    double m = 0;
    for (int i = threadIdx.x; i < p_pad; i += blockDim.x) {
        double d;
        if (i < p) {
            d = G[size_t(i) * ld + i] + ridge;
        } else {
            d = 1.0;
        }
        G[size_t(i) * ld + i] = d;
        dg0[i] = d;
        if (i < p) m = fmax(m, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double mm = 0;
        for (int i = 0; i < int(blockDim.x >> 6); ++i) mm = fmax(mm, red[i]);
        gmax[0] = mm;
    }
}

__global__ void __launch_bounds__(RT) k_add_diag_scaled(double *__restrict__ G, int ld, int p,
                                                        const double *__restrict__ gmax, double rel,
                                                        double *__restrict__ dg0, int *__restrict__ info, int n_info) {
    const int i = blockIdx.x * RT + threadIdx.x;
    for (int e = i; e < n_info; e += int(gridDim.x) * RT) info[e] = 0;
    if (i >= p) return;
    const double d = G[size_t(i) * ld + i] + rel * gmax[0];
    G[size_t(i) * ld + i] = d;
    dg0[i] = d;
}

// ---- factorisation: chol_step.hip -------------------------------------------------------------------
// info of a factorisation: cp_common.h
constexpr int chol_info_count(int nblk) { return cp_chol_info_count(nblk); }
typedef double v4f64c __attribute__((ext_vector_type(4)));
constexpr int CP_REFIT_MAX_BATCH = 16;

__global__ void __launch_bounds__(RT) k_axpy(double *__restrict__ y, const double *__restrict__ x, size_t count) {
    size_t i = blockIdx.x * size_t(RT) + threadIdx.x;
    const size_t step = size_t(gridDim.x) * RT;
    for (; i < count; i += step) y[i] += x[i];
}

// coef[j, col] = W[col, j];  b[j] = ymean[j] - sum_col xmean[col] coef[j, col]
// coef_host / b_host / info_host: optional copies in pinned host memory the device writes directly
// (no copy packets after the last kernel); info_host[0] = first failed pivot + 1 of the factorisation.
__global__ void __launch_bounds__(RT) k_finalize(const double *__restrict__ W, int ldw, int p, int n,
                                                 const double *__restrict__ xmean, const double *__restrict__ ymean,
                                                 double *__restrict__ coef, double *__restrict__ b,
                                                 double *__restrict__ coef_host, double *__restrict__ b_host,
                                                 const int *__restrict__ info, int *__restrict__ info_host) {
    __shared__ double red[RT / 64];
    const int j = blockIdx.x;
    if (j == 0 && threadIdx.x == 0) info_host[0] = info[0];
    double s = 0;
    for (int col = threadIdx.x; col < p; col += RT) {
        const double v = W[size_t(col) * ldw + j];
        coef[size_t(j) * p + col] = v;
        if (coef_host) coef_host[size_t(j) * p + col] = v;
        s += xmean[col] * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int i = 0; i < RT / 64; ++i) tot += red[i];
        b[j] = ymean[j] - tot;
        if (b_host) b_host[j] = ymean[j] - tot;
    }
    (void)n;
}

// The same for large (p, n): 32 x 32 tiles through LDS so that both the reads of W (rows of n_pad) and the writes of coef
// (rows of p, also into the pinned host copy) are coalesced; the intercept from per-tile partial sums, fixed order.
__global__ void __launch_bounds__(RT) k_finalize_tile(const double *__restrict__ W, int ldw, int p, int n,
                                                      const double *__restrict__ xmean, double *__restrict__ coef,
                                                      double *__restrict__ coef_host, double *__restrict__ part, int ld_part) {
    __shared__ double t[32][33];
    __shared__ double red[8][33];
    const int col0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double sacc = 0.0;
    for (int y = ty; y < 32; y += 8) {
        const int col = col0 + y;
        const double v = (col < p && j0 + tx < n) ? W[size_t(col) * ldw + j0 + tx] : 0.0;
        t[y][tx] = v;
        if (col < p) sacc = fma(xmean[col], v, sacc);
    }
    red[ty][tx] = sacc;
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int j = j0 + y, col = col0 + tx;
        if (j < n && col < p) {
            const double v = t[tx][y];
            coef[size_t(j) * p + col] = v;
            if (coef_host) coef_host[size_t(j) * p + col] = v;
        }
    }
    if (ty == 0 && j0 + tx < n) {
        double tot = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) tot += red[r][tx];
        part[size_t(blockIdx.x) * ld_part + j0 + tx] = tot;
    }
}
__global__ void __launch_bounds__(RT) k_finalize_bias(const double *__restrict__ part, int nparts, int ld_part, int n,
                                                      const double *__restrict__ ymean, double *__restrict__ b,
                                                      double *__restrict__ b_host, const int *__restrict__ info,
                                                      int *__restrict__ info_host) {
    const int j = blockIdx.x * RT + threadIdx.x;
    if (j == 0) info_host[0] = info[0];
    if (j >= n) return;
    double tot = 0;
    for (int r = 0; r < nparts; ++r) tot += part[size_t(r) * ld_part + j];
    b[j] = ymean[j] - tot;
    if (b_host) b_host[j] = ymean[j] - tot;
}
// W [p_pad, n_pad] -> coef / b (+ pinned host copies): tiled for large outputs (part: scratch of (p_pad / 32) * n_pad doubles)
int finalize_launch(cp_ctx *ctx, const double *W, int n_pad, int p, int n, const double *xmean, const double *ymean,
                    double *coef, double *b, double *coef_host, double *b_host, const int *info, int *info_host,
                    double *part) {
    if (part && size_t(p) * n >= (1u << 18)) {
        const int gx = (p + 31) / 32, gy = (n + 31) / 32;
        k_finalize_tile<<<dim3(gx, gy), RT, 0, ctx->stream>>>(W, n_pad, p, n, xmean, coef, coef_host, part, n_pad);
        CP_LAUNCH_CHECK(ctx);
        k_finalize_bias<<<(n + RT - 1) / RT, RT, 0, ctx->stream>>>(part, gx, n_pad, n, ymean, b, b_host, info, info_host);
        CP_LAUNCH_CHECK(ctx);
    } else {
        k_finalize<<<n, RT, 0, ctx->stream>>>(W, n_pad, p, n, xmean, ymean, coef, b, coef_host, b_host, info, info_host);
        CP_LAUNCH_CHECK(ctx);
    }
    return CP_OK;
}

struct Chol {
    double *G;   // p_pad x p_pad working matrix (trailing Schur complements)
    double *U;   // the factor (upper), written block row by block row (never in place: the panel
                 // products then use 64x64 tiles and spread over 4x more CUs)
    double *Lt;  // U^T
    double *TI, *TIT;
    double *dg0, *gmax;
    int *info;
    int p, p_pad, nblk;
};

// G = U^T U (upper, into ch.U), TI / TIT per diagonal block, the off-diagonal blocks of Lt = U^T: one launch per
// 128-column step (chol_step.hip).  fwd_R (optional, p_pad x fwd_n_pad): right-hand sides whose forward substitution
// U^T y = r rides in those launches; *fwd_done tells the caller that only the backward sweep is left.
int chol_factor(cp_ctx *ctx, Chol &ch, double piv_tol, double *fwd_R = nullptr, int fwd_n_pad = 0, bool *fwd_done = nullptr) {
    const bool fwd = fwd_R && fwd_n_pad > 0 && fwd_n_pad % NB == 0;
    if (fwd_done) *fwd_done = fwd;
    return cp_chol_factor_steps(ctx, ch.G, ch.U, ch.Lt, ch.p_pad, ch.nblk, ch.dg0, piv_tol, ch.TI, ch.TIT, ch.info,
                                fwd ? fwd_R : nullptr, fwd ? fwd_n_pad : 0);
}

// Rm <- (U^T U)^-1 Rm (p_pad x n_pad), in place, ONE launch.  Triangular solves are independent per
// right-hand-side column: workgroup g owns columns [16 g, 16 g + 16) and runs the whole forward
// (U^T y = r) and backward (U w = y) block substitution for them -- no inter-workgroup traffic, no
// per-block launches (the launch-per-block version was 40 dependent GEMM launches, 0.7 ms alone and
// 3 ms with other layers in flight).  Per 128-row block b:
//   S   = R_b - sum_{k<b} U[k,b]^T Y_k      wave m accumulates its 16 x 16 row tile on MFMA (K = 128 b)
//   Y_b = U_bb^-T S                          through LDS, with the inverted diagonal block TI_b
// and the mirror image with Lt = U^T / TIT_b for the backward sweep.  8 waves = 8 row tiles.
struct StripFinal {  // optional tail of k_solve_strips: what k_finalize does, for the strip's 16 columns
    int p, n;                      // p == 0: no tail
    const double *xmean, *ymean;
    double *coef, *b;              // device outputs: coef[n, p], b[n]
    double *coef_host, *b_host;    // pinned host copies (may be null)
    const int *info;
    int *info_host;
    // Banded backward substitution (chol_solve_blocked): the rows of W become final band by band, from the bottom up, and
    // leave for the host AS THEY DO -- the launch of a band first lays out the rows the PREVIOUS band finished (rows
    // [pre_lo, pre_hi): posted writes over PCIe that drain while this band substitutes), the launch of the top band also its
    // own (rows [post_lo, post_hi)) and the intercept.  b accumulates sum_col xmean[col] W[col, j] across the launches (first:
    // the launch that starts it; last: the one that turns it into ymean - sum).  All four bounds 0: the whole-matrix tail above.
    int pre_lo = 0, pre_hi = 0, post_lo = 0, post_hi = 0;
    int first = 0, last = 0;
};

// rows [lo, hi) of W (already final) for this strip's 16 right-hand sides: coef[j, col] = W[col, j] (device + pinned host copy),
// their share of sum_col xmean[col] W[col, j] added to b[j] (start: written instead); finish: b[j] = ymean[j] - sum
__device__ __forceinline__ void strip_finalize_rows(const double *R, int n_pad, const StripFinal &fin, int col0, int lo, int hi,
                                                    bool start, bool finish, double *red /* 8 x 16 doubles of LDS */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (hi > fin.p) hi = fin.p;
    double acc16[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) acc16[jj] = 0.0;
    for (int col = lo + int(threadIdx.x); col < hi; col += 512) {
        const double xm = fin.xmean[col];
        const double *wr = R + size_t(col) * n_pad + col0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = col0 + jj;
            if (j < fin.n) {
                const double v = wr[jj];
                fin.coef[size_t(j) * fin.p + col] = v;
                if (fin.coef_host) fin.coef_host[size_t(j) * fin.p + col] = v;
                acc16[jj] = fma(xm, v, acc16[jj]);
            }
        }
    }
    __syncthreads();      // `red` may alias the substitution's LDS
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        double v = acc16[jj];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave * 16 + jj] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16 && col0 + int(threadIdx.x) < fin.n) {
        double tot = 0;
        for (int w8 = 0; w8 < 8; ++w8) tot += red[w8 * 16 + threadIdx.x];
        const int j = col0 + threadIdx.x;
        if (!start) tot += fin.b[j];
        const double bj = finish ? fin.ymean[j] - tot : tot;
        fin.b[j] = bj;
        if (finish && fin.b_host) fin.b_host[j] = bj;
    }
    __syncthreads();
}

// SWEEPS: bit 0 = forward (U^T y = r), bit 1 = backward (U w = y); 3 = both (the normal-equation solve)
template <int SWEEPS = 3>
__device__ __forceinline__ void solve_strips_body(const double *__restrict__ U, const double *__restrict__ Lt, int ld,
                                                  const double *__restrict__ TI, const double *__restrict__ TIT,
                                                  int nblk, double *R, int n_pad, const StripFinal &fin, int b0 = 0,
                                                  int b1 = -1) {
    // [b0, b1): the band of 128-row blocks this launch substitutes through (default: all of them).  A band launch only
    // accounts for the couplings INSIDE the band; chol_solve_blocked applies the others as chip-filling GEMMs.
    if (b1 < 0) b1 = nblk;
    __shared__ double S[NB][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fk = lane >> 4, fi = lane & 15;
    const int col0 = blockIdx.x * 16, row0 = wave * 16;
    const bool banded_tail = fin.p > 0 && (fin.pre_hi > fin.pre_lo || fin.post_hi > fin.post_lo);
    if (banded_tail && fin.pre_hi > fin.pre_lo)     // the rows the previous band finished: on their way while this band works
        strip_finalize_rows(R, n_pad, fin, col0, fin.pre_lo, fin.pre_hi, fin.first != 0, false, &S[0][0]);
    for (int sweep = 0; sweep < 2; ++sweep) {
        if (!(SWEEPS & (1 << sweep))) continue;
        const double *Tri = sweep == 0 ? U : Lt;      // element (kk of block k, m of block b) at Tri[(k NB + kk) ld + b NB + m]
        const double *Dinv = sweep == 0 ? TI : TIT;   // a-operand of the diagonal solve: Dinv_b[j, m]
        for (int step = 0; step < b1 - b0; ++step) {
            const int b = sweep == 0 ? b0 + step : b1 - 1 - step;
            double *Rb = R + (size_t(b) * NB + row0) * n_pad + col0 + fi;
            v4f64c acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = Rb[size_t(fk + 4 * r) * n_pad];
            const int k_lo = sweep == 0 ? b0 : b + 1, k_hi = sweep == 0 ? b : b1;
            for (int k = k_lo; k < k_hi; ++k) {
                const double *Ak = Tri + size_t(k) * NB * ld + size_t(b) * NB + row0 + fi;
                const double *Yk = R + size_t(k) * NB * n_pad + col0 + fi;
#pragma unroll 8
                for (int q = 0; q < NB / 4; ++q) {
                    const double av = -Ak[size_t(4 * q + fk) * ld];
                    const double bv = Yk[size_t(4 * q + fk) * n_pad];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) S[row0 + fk + 4 * r][fi] = acc[r];
            __syncthreads();
            const double *Db = Dinv + size_t(b) * NB * NB + row0 + fi;
            v4f64c o = {0., 0., 0., 0.};
#pragma unroll 8
            for (int q = 0; q < NB / 4; ++q)
                o = __builtin_amdgcn_mfma_f64_16x16x4f64(Db[size_t(4 * q + fk) * NB], S[4 * q + fk][fi], o, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Rb[size_t(fk + 4 * r) * n_pad] = o[r];
            __syncthreads();  // block b of the strip is complete (and visible) before any wave reads it
        }
    }
    if (banded_tail) {
        if (fin.post_hi > fin.post_lo) {
            if (blockIdx.x == 0 && threadIdx.x == 0 && fin.info_host && fin.last) fin.info_host[0] = fin.info[0];
            strip_finalize_rows(R, n_pad, fin, col0, fin.post_lo, fin.post_hi, fin.first != 0 && !(fin.pre_hi > fin.pre_lo),
                                fin.last != 0, &S[0][0]);
        }
        return;
    }
    if (fin.p > 0) {  // coef[j, col] = W[col, j],  b[j] = ymean[j] - sum_col xmean[col] W[col, j]  for this strip's j
        if (blockIdx.x == 0 && threadIdx.x == 0 && fin.info_host) fin.info_host[0] = fin.info[0];
        double acc16[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) acc16[jj] = 0.0;
        for (int col = threadIdx.x; col < fin.p; col += 512) {
            const double xm = fin.xmean[col];
            const double *wr = R + size_t(col) * n_pad + col0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = col0 + jj;
                if (j < fin.n) {
                    const double v = wr[jj];
                    fin.coef[size_t(j) * fin.p + col] = v;
                    if (fin.coef_host) fin.coef_host[size_t(j) * fin.p + col] = v;
                    acc16[jj] = fma(xm, v, acc16[jj]);
                }
            }
        }
        double *red = &S[0][0];  // 8 waves x 16 partial sums
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            double v = acc16[jj];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[wave * 16 + jj] = v;
        }
        __syncthreads();
        if (threadIdx.x < 16 && col0 + int(threadIdx.x) < fin.n) {
            double tot = 0;
            for (int w8 = 0; w8 < 8; ++w8) tot += red[w8 * 16 + threadIdx.x];
            const int j = col0 + threadIdx.x;
            const double bj = fin.ymean[j] - tot;
            fin.b[j] = bj;
            if (fin.b_host) fin.b_host[j] = bj;
        }
    }
}

template <int SWEEPS>
__global__ void __launch_bounds__(512) k_solve_strips(const double *__restrict__ U, const double *__restrict__ Lt, int ld,
                                                      const double *__restrict__ TI, const double *__restrict__ TIT,
                                                      int nblk, double *R, int n_pad, StripFinal fin, int b0 = 0, int b1 = -1) {
    solve_strips_body<SWEEPS>(U, Lt, ld, TI, TIT, nblk, R, n_pad, fin, b0, b1);
}

struct StripJob {
    const double *U, *Lt;
    int ld;
    const double *TI, *TIT;
    int nblk;
    double *R;
    int n_pad;
    StripFinal fin;
};
struct StripBatch {
    StripJob j[CP_REFIT_MAX_BATCH];
};
template <int SWEEPS>
__global__ void __launch_bounds__(512) k_solve_strips_batch(StripBatch b) {  // blockIdx.y = job
    const StripJob &a = b.j[blockIdx.y];
    if (int(blockIdx.x) * 16 >= a.n_pad) return;
    solve_strips_body<SWEEPS>(a.U, a.Lt, a.ld, a.TI, a.TIT, a.nblk, a.R, a.n_pad, a.fin);
}

int chol_solve(cp_ctx *ctx, const Chol &ch, double *Rm, double *, int n_pad, const StripFinal &fin = StripFinal{},
               int sweeps = 3) {
    if (sweeps == 1)
        k_solve_strips<1><<<n_pad / 16, 512, 0, ctx->stream>>>(ch.U, ch.Lt, ch.p_pad, ch.TI, ch.TIT, ch.nblk, Rm, n_pad, fin);
    else if (sweeps == 2)
        k_solve_strips<2><<<n_pad / 16, 512, 0, ctx->stream>>>(ch.U, ch.Lt, ch.p_pad, ch.TI, ch.TIT, ch.nblk, Rm, n_pad, fin);
    else
        k_solve_strips<3><<<n_pad / 16, 512, 0, ctx->stream>>>(ch.U, ch.Lt, ch.p_pad, ch.TI, ch.TIT, ch.nblk, Rm, n_pad, fin);
    CP_LAUNCH_CHECK(ctx);
    return CP_OK;
}

// Both substitutions for LARGE factors.  The one-launch strips above read the whole factor once per 16 right-hand
// sides (n/16 x p^2/2 x 8 B per sweep: 4.8 GB at p = 4350, n = 512 -- load bound on 32 workgroups, 4.6 ms).  Here the
// rows are processed in bands of OB = 4 blocks: inside a band the strips kernel (short: 4 block steps), and the
// band's contribution to every remaining row as ONE chip-filling f64 MFMA GEMM with K = 512
//   forward   R[rows below] -= U[band, below]^T Y[band]       backward   R[rows above] -= (Lt[band, above])^T W[band]
// so the factor is read once per sweep.  The coefficient lay-out / intercept tail is the caller's (k_finalize).
constexpr int SOLVE_OB = 4;   // blocks per band (8: measured equal in the vgg16 job, 27.8 vs 27.6 ms)
constexpr int solve_blocked_min_blocks() { return 16; }   // banded substitution from 16 blocks (p > 1920) on
size_t chol_solve_blocked_workspace(const cp_ctx *ctx, int p_pad, int n_pad) {
    return cp_gemm_tn_workspace(ctx, p_pad, n_pad, SOLVE_OB * NB, CP_TRI_NONE);
}
// fin (optional): the coefficient lay-out / intercept ride in the launches of the BACKWARD sweep, band by band (StripFinal);
// *fin_done tells the caller that nothing is left to finalize.
int chol_solve_blocked(cp_ctx *ctx, const Chol &ch, double *Rm, int n_pad, int sweeps = 3, const StripFinal *fin = nullptr,
                       bool *fin_done = nullptr) {
    const int ld = ch.p_pad, nblk = ch.nblk;
    const StripFinal none{};
    if (fin_done) *fin_done = false;
    if (sweeps & 1) {
        for (int b0 = 0; b0 < nblk; b0 += SOLVE_OB) {
            const int b1 = std::min(nblk, b0 + SOLVE_OB);
            k_solve_strips<1><<<n_pad / 16, 512, 0, ctx->stream>>>(ch.U, ch.Lt, ld, ch.TI, ch.TIT, nblk, Rm, n_pad, none, b0, b1);
            CP_LAUNCH_CHECK(ctx);
            const int below = (nblk - b1) * NB;
            if (below > 0)
                CP_TRY(cp_gemm_tn_f64(ctx, below, n_pad, (b1 - b0) * NB, -1.0, ch.U + size_t(b0) * NB * ld + size_t(b1) * NB, ld,
                                      Rm + size_t(b0) * NB * n_pad, n_pad, 1.0, Rm + size_t(b1) * NB * n_pad, n_pad, CP_TRI_NONE));
        }
    }
    if (sweeps & 2) {
        for (int b1 = nblk; b1 > 0; b1 -= SOLVE_OB) {
            const int b0 = std::max(0, b1 - SOLVE_OB);
            StripFinal f = none;
            if (fin && fin->p > 0) {
                f = *fin;
                f.first = b1 == nblk ? 0 : (b1 + SOLVE_OB >= nblk ? 1 : 0);   // the launch that lays out the first rows starts b
                if (b1 < nblk) {                                          // the rows of the band above... below: finished by the previous launch
                    f.pre_lo = b1 * NB;
                    f.pre_hi = std::min(nblk, b1 + SOLVE_OB) * NB;
                }
                if (b0 == 0) {                                            // the top band: its own rows too, and the intercept
                    f.post_lo = 0;
                    f.post_hi = b1 * NB;
                    f.last = 1;
                    if (b1 == nblk) f.first = 1;                          // a single band: everything in this launch
                }
                if (f.pre_hi <= f.pre_lo && f.post_hi <= f.post_lo) f = none;   // the first of several bands: nothing final yet
            }
            k_solve_strips<2><<<n_pad / 16, 512, 0, ctx->stream>>>(ch.U, ch.Lt, ld, ch.TI, ch.TIT, nblk, Rm, n_pad, f, b0, b1);
            CP_LAUNCH_CHECK(ctx);
            const int above = b0 * NB;
            if (above > 0)
                CP_TRY(cp_gemm_tn_f64(ctx, above, n_pad, (b1 - b0) * NB, -1.0, ch.Lt + size_t(b0) * NB * ld, ld,
                                      Rm + size_t(b0) * NB * n_pad, n_pad, 1.0, Rm, n_pad, CP_TRI_NONE));
        }
        if (fin && fin->p > 0 && fin_done) *fin_done = true;
    }
    return CP_OK;
}

constexpr double PIV_TOL = 1e-6;   // normal equations are trusted while pivot / original diagonal stays above this

// ---- helpers of the rank-revealing path --------------------------------------------------------------------
// gn[0] = max_i sum_j |G[i,j]| over the leading p x p part (one workgroup per row, fixed-order combine by the caller's
// second launch with rows = 1)
__global__ void __launch_bounds__(RT) k_abs_row_sums(const double *__restrict__ G, int ld, int p, double *__restrict__ out) {
    __shared__ double red[RT / 64];
    const int i = blockIdx.x;
    double sacc = 0;
    for (int j = threadIdx.x; j < p; j += RT) sacc += fabs(G[size_t(i) * ld + j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sacc;
    __syncthreads();
    if (threadIdx.x == 0) out[i] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(RT) k_max_of(const double *__restrict__ v, int count, double *__restrict__ out) {
    __shared__ double red[RT / 64];
    double m = 0;
    for (int i = threadIdx.x; i < count; i += RT) m = fmax(m, v[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
// dst[c, r] = src[r, c] for r < rows, c < cols, zero elsewhere in the dst_rows x dst_cols (both % 32 == 0) target
__global__ void __launch_bounds__(RT) k_transpose_pad(const double *__restrict__ src, int rows, int cols, int ld_src,
                                                      double *__restrict__ dst, int ld_dst) {
    __shared__ double t[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;   // tile of src
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) t[y][tx] = (r0 + y < rows && c0 + tx < cols) ? src[size_t(r0 + y) * ld_src + c0 + tx] : 0.0;
    __syncthreads();
    for (int y = ty; y < 32; y += 8) dst[size_t(c0 + y) * ld_dst + r0 + tx] = t[tx][y];
}
// ref[i] = max(ref[i], rel * gmax[0]): a column of the preconditioned design that is pure rounding noise (an exact copy
// of earlier columns: norm^2 ~ eps^2 / shift ~ 1e-20 of the largest) must not pass the RELATIVE pivot test
__global__ void __launch_bounds__(RT) k_floor_ref(double *__restrict__ ref, int p, const double *__restrict__ gmax, double rel) {
    const int i = blockIdx.x * RT + threadIdx.x;
    if (i < p) ref[i] = fmax(ref[i], rel * gmax[0]);
}
// M[i, :] *= f(scale[i]) for i < rows; rows in [rows, rows_total) are zeroed.  mode 0: * s, 1: / s, 2: / s^2
__global__ void __launch_bounds__(RT) k_scale_rows(double *__restrict__ M, int ld, int rows, int cols,
                                                   const double *__restrict__ scale, int mode) {
    const int i = blockIdx.x;
    double f = 0.0;
    if (i < rows) {
        const double sv = scale[i];
        f = mode == 0 ? sv : (mode == 1 ? 1.0 / sv : 1.0 / (sv * sv));
    }
    for (int j = threadIdx.x; j < cols; j += RT) M[size_t(i) * ld + j] = i < rows ? M[size_t(i) * ld + j] * f : 0.0;
}

}  // namespace

extern "C" int cp_debug_itq_sweeps(cp_ctx *ctx) { return ctx ? ctx->itq_sweeps : -1; }
// 1: the last refit of this context formed G and X^T Y in ONE launch (cp_gemm_gram_xty) -- its "refit_gram_gemm" bracket holds both
extern "C" int cp_debug_last_xty_fused(cp_ctx *ctx) { return ctx ? int(ctx->last_xty_fused) : -1; }
// what: 0 = Newton-Schulz steps, 1 = alternations that took the sign-function route (of the last cp_itq_iterate)
extern "C" int cp_debug_itq_sign(cp_ctx *ctx, int what) {
    return !ctx ? -1 : (what == 0 ? ctx->itq_ns_steps : ctx->itq_sign_alternations);
}

namespace {

// Everything after the normal equations exist: factor, substitute, lay out (W, b); shared by the single-GPU refit
// and by the sample-sharded one (whose Gram arrives from an all-reduce instead of from this rank's GEMM).
struct RefitSolve {
    double *G, *G0, *Lt, *Uf, *Yt, *Rm, *R2, *TI, *TIT, *xmean, *ymean, *dg0, *gmax;
    int *dinfo;
    int p, p_pad, n, n_pad, nblk;
    int64_t N;  // rows behind the Gram (all ranks)
    double ridge;
    double *W_out, *b_out;
    int *info_host;
    double *b_host, *W_host;
    // the centred rows themselves (null in the row-sharded tail, which only sees the reduced Gram)
    const double *Xs = nullptr, *Yc = nullptr;
    int64_t N_pad = 0;
};

// scratch of the rank-revealing path: its own device allocation swapped in as the context's arena for the duration
// (the regular arena keeps Xs / Yc / means alive; this path is rare, so a hipMalloc per call is fine)
struct ArenaSwap {
    cp_ctx *ctx;
    char *old_arena;
    size_t old_bytes, old_used;
    char *mine = nullptr;
    explicit ArenaSwap(cp_ctx *c) : ctx(c), old_arena(c->arena), old_bytes(c->arena_bytes), old_used(c->arena_used) {}
    int open(size_t bytes) {
        bytes = cp_align_up(bytes + (1 << 20), 1 << 20);
        if (hipMalloc(reinterpret_cast<void **>(&mine), bytes) != hipSuccess)
            return cp_set_error(ctx, CP_ERR_NOMEM, "refit (rank-revealing path): hipMalloc(%zu)", bytes);
        ctx->arena = mine;
        ctx->arena_bytes = bytes;
        ctx->arena_used = 0;
        return CP_OK;
    }
    ~ArenaSwap() {
        if (mine) {
            (void)hipStreamSynchronize(ctx->stream);
            ctx->arena = old_arena;
            ctx->arena_bytes = old_bytes;
            ctx->arena_used = old_used;
            (void)hipFree(mine);
        }
    }
};

int read_back(cp_ctx *ctx, void *host, const void *dev, size_t bytes) {
    CP_HIP(ctx, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CP_HIP(ctx, cp_stream_wait(ctx));
    return CP_OK;
}

// Least squares as accurate as gelsd on ill-conditioned and rank-deficient designs (see the header of this file).
// G0: pristine Gram buffer (overwritten), everything else of `rs` is reused as work space; outputs through finalize().
template <class NormalEquations, class Finalize>
int refit_robust(cp_ctx *ctx, const RefitSolve &rs, NormalEquations &&normal_equations, Finalize &&finalize,
                 cp_refit_info *info) {
    const int p = rs.p, p_pad = rs.p_pad, n_pad = rs.n_pad, nblk = rs.nblk;
    const int64_t N = rs.N, N_pad = rs.N_pad;
    const int Nr = int(cp_align_up(size_t(N_pad), 128));
    const double eps = 2.220446049250313e-16;
    const size_t g_c = size_t(p_pad) * p_pad, r_c = size_t(p_pad) * n_pad, x_c = size_t(Nr) * p_pad,
                 ti_c = size_t(nblk) * NB * NB;
    // pristine Gram (G0) and right-hand side (R2) from the regular arena; the diagonal is prepared (pad = 1, dg0, gmax)
    CP_TRY(normal_equations(rs.G0, rs.R2, false));
    size_t ws = std::max(cp_gemm_tn_workspace(ctx, p_pad, p_pad, int(N_pad), CP_TRI_LOWER_MIRROR),
                         cp_gemm_tn_workspace(ctx, p_pad, n_pad, int(N_pad), CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, p_pad, p_pad, p_pad, CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, p_pad, n_pad, p_pad, CP_TRI_NONE));
    ArenaSwap tmp(ctx);
    CP_TRY(tmp.open((2 * x_c + 9 * g_c + 3 * r_c + 2 * ti_c + 8 * size_t(p_pad)) * 8 + 2 * SvdScratch::bytes(p_pad, p_pad) +
                    size_t(chol_info_count(nblk) + 64) * 4 + ws + (1 << 20)));
    double *XsT = cp_arena_take_t<double>(ctx, x_c), *X1 = cp_arena_take_t<double>(ctx, x_c);
    double *G2 = cp_arena_take_t<double>(ctx, g_c), *Gw = cp_arena_take_t<double>(ctx, g_c);
    double *U2 = cp_arena_take_t<double>(ctx, g_c), *Lt2 = cp_arena_take_t<double>(ctx, g_c);
    double *TI2 = cp_arena_take_t<double>(ctx, ti_c), *TIT2 = cp_arena_take_t<double>(ctx, ti_c);
    double *rowsum = cp_arena_take_t<double>(ctx, p_pad), *gnorm = cp_arena_take_t<double>(ctx, 8);
    double *dg2 = cp_arena_take_t<double>(ctx, p_pad), *gmax2 = cp_arena_take_t<double>(ctx, 8);
    int *dinfo2 = cp_arena_take_t<int>(ctx, chol_info_count(nblk) + 16);
    if (!XsT || !X1 || !G2 || !Gw || !U2 || !Lt2 || !TI2 || !TIT2 || !rowsum || !gnorm || !dg2 || !gmax2 || !dinfo2)
        return cp_set_error(ctx, CP_ERR_NOMEM, "refit (rank-revealing path): scratch");

    // ---- stage 1: R1 = chol(G + s I) into (Uf, Lt, TI, TIT) ----
    k_abs_row_sums<<<p, RT, 0, ctx->stream>>>(rs.G0, p_pad, p, rowsum);
    CP_LAUNCH_CHECK(ctx);
    k_max_of<<<1, RT, 0, ctx->stream>>>(rowsum, p, gnorm);
    CP_LAUNCH_CHECK(ctx);
    Chol ch1{rs.G, rs.Uf, rs.Lt, rs.TI, rs.TIT, rs.dg0, rs.gmax, rs.dinfo, p, p_pad, nblk};
    double rel = 4.0 * double(std::max<int64_t>(N, p)) * eps;
    int h = 0;
    for (int attempt = 0; attempt < 4; ++attempt, rel *= 100.0) {
        CP_HIP(ctx, hipMemcpyAsync(rs.G, rs.G0, g_c * 8, hipMemcpyDeviceToDevice, ctx->stream));
        CP_HIP(ctx, hipMemsetAsync(rs.Uf, 0, g_c * 8, ctx->stream));   // B = T V^T R1 below reads the whole of R1
        k_add_diag_scaled<<<(p + RT - 1) / RT, RT, 0, ctx->stream>>>(rs.G, p_pad, p, gnorm, rel, rs.dg0, rs.dinfo,
                                                                    chol_info_count(nblk));
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(chol_factor(ctx, ch1, 0.0));
        CP_TRY(read_back(ctx, &h, rs.dinfo, sizeof(int)));
        if (h == 0) break;
    }
    if (h != 0) return cp_set_error(ctx, CP_ERR_NUMERIC, "refit: shifted factorisation broke down at column %d", h - 1);
    // X1 = Xs R1^-1: forward substitution U1^T X1^T = Xs^T with the N rows as right-hand sides
    k_transpose_pad<<<dim3(Nr / 32, p_pad / 32), RT, 0, ctx->stream>>>(rs.Xs, int(N_pad), p_pad, p_pad, XsT, Nr);
    CP_LAUNCH_CHECK(ctx);
    CP_TRY(chol_solve(ctx, ch1, XsT, nullptr, Nr, StripFinal{}, 1));
    k_transpose_pad<<<dim3(p_pad / 32, Nr / 32), RT, 0, ctx->stream>>>(XsT, p_pad, Nr, Nr, X1, p_pad);
    CP_LAUNCH_CHECK(ctx);
    // ---- stage 2: normal equations of the preconditioned rows ----
    CP_TRY(cp_gemm_tn_f64(ctx, p_pad, p_pad, int(N_pad), 1.0, X1, p_pad, X1, p_pad, 0.0, G2, p_pad, CP_TRI_LOWER_MIRROR));
    CP_TRY(cp_gemm_tn_f64(ctx, p_pad, n_pad, int(N_pad), 1.0, X1, p_pad, rs.Yc, n_pad, 0.0, rs.R2, n_pad, CP_TRI_NONE));
    CP_HIP(ctx, hipMemcpyAsync(Gw, G2, g_c * 8, hipMemcpyDeviceToDevice, ctx->stream));
    k_diag_prepare<<<1, 256, 0, ctx->stream>>>(Gw, p_pad, p, p_pad, 0.0, dg2, gmax2, dinfo2, chol_info_count(nblk));
    CP_LAUNCH_CHECK(ctx);
    k_floor_ref<<<(p + RT - 1) / RT, RT, 0, ctx->stream>>>(dg2, p, gmax2, 1e-5);   // pivot <= max(1e-11 diag, 1e-16 max diag)
    CP_LAUNCH_CHECK(ctx);
    Chol ch2{Gw, U2, Lt2, TI2, TIT2, dg2, gmax2, dinfo2, p, p_pad, nblk};
    CP_TRY(chol_factor(ctx, ch2, 1e-11));
    CP_TRY(read_back(ctx, &h, dinfo2, sizeof(int)));
    if (h == 0) {   // path 1: full column rank
        CP_HIP(ctx, hipMemcpyAsync(rs.Rm, rs.R2, r_c * 8, hipMemcpyDeviceToDevice, ctx->stream));
        CP_TRY(chol_solve(ctx, ch2, rs.Rm, nullptr, n_pad));                       // Z = G2^-1 C2
        CP_TRY(chol_solve(ctx, ch1, rs.Rm, nullptr, n_pad, StripFinal{}, 2));      // W = R1^-1 Z
        cp_stage_mark(ctx, "refit_robust_cholqr");
        CP_HIP(ctx, hipMemsetAsync(rs.dinfo, 0, sizeof(int), ctx->stream));
        CP_TRY(finalize());
        info->p = p;
        info->rank = p;
        info->fallback = 2;
        info->reserved = 0;
        return CP_OK;
    }
    // ---- path 2: rank revealing.  G2 = V T^2 V^T ----
    const int p_e = p;   // rows taking part in the decompositions
    double *lam = cp_arena_take_t<double>(ctx, p_pad), *Vt = cp_arena_take_t<double>(ctx, g_c);
    double *SH = cp_arena_take_t<double>(ctx, g_c);
    SvdScratch sc;
    if (!lam || !Vt || !SH || !sc.take(ctx, p_pad, p_pad))
        return cp_set_error(ctx, CP_ERR_NOMEM, "refit (rank-revealing path): scratch (decomposition)");
    CP_HIP(ctx, hipMemsetAsync(Vt, 0, g_c * 8, ctx->stream));
    CP_HIP(ctx, hipMemsetAsync(SH, 0, g_c * 8, ctx->stream));
    // eigenvalues below 10 p eps are dropped below anyway: rows at that level are noise and only keep the sweeps going
    CP_TRY(cp_svd_rows_core(ctx, G2, p_pad, p_e, p_pad, p_e, lam, Vt, p_pad, SH, p_pad, sc, nullptr, false, 1e-14, 0.0));
    std::vector<double> hl(p_e);
    CP_TRY(read_back(ctx, hl.data(), lam, size_t(p_e) * 8));
    int r = 0;
    while (r < p_e && hl[r] > 10.0 * double(p) * eps * hl[0] && hl[r] > 0.0) ++r;
    int k = 0;
    if (r > 0) {
        std::vector<double> tau(r);
        for (int i = 0; i < r; ++i) tau[i] = std::sqrt(hl[i]);
        double *dtau = lam;   // reuse: tau_i on the device
        CP_HIP(ctx, hipMemcpyAsync(dtau, tau.data(), size_t(r) * 8, hipMemcpyHostToDevice, ctx->stream));
        const int r_pad = int(cp_align_up(size_t(r), 128));
        // V_L as a [p_pad, r_pad] k-major operand, B = T V_L^T R1 (r x p), C = T^-1 V_L^T C2 (r x n)  [Gw, U2 reused]
        double *Vtr = Gw, *B = U2, *Cm = Lt2;   // r_pad x p_pad <= p_pad x p_pad each; Cm: r_pad x n_pad <= needs r_c
        double *Cbuf = cp_arena_take_t<double>(ctx, std::max(r_c, g_c)), *Dm = cp_arena_take_t<double>(ctx, std::max(r_c, g_c));
        if (!Cbuf || !Dm) return cp_set_error(ctx, CP_ERR_NOMEM, "refit (rank-revealing path): scratch (products)");
        Cm = Cbuf;
        k_transpose_pad<<<dim3(r_pad / 32, p_pad / 32), RT, 0, ctx->stream>>>(Vt, r, p_pad, p_pad, Vtr, r_pad);
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(cp_gemm_tn_f64(ctx, r_pad, p_pad, p_pad, 1.0, Vtr, r_pad, rs.Uf, p_pad, 0.0, B, p_pad, CP_TRI_NONE));
        k_scale_rows<<<r_pad, RT, 0, ctx->stream>>>(B, p_pad, r, p_pad, dtau, 0);
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(cp_gemm_tn_f64(ctx, r_pad, n_pad, p_pad, 1.0, Vtr, r_pad, rs.R2, n_pad, 0.0, Cm, n_pad, CP_TRI_NONE));
        k_scale_rows<<<r_pad, RT, 0, ctx->stream>>>(Cm, n_pad, r, n_pad, dtau, 1);
        CP_LAUNCH_CHECK(ctx);
        // B = Ub S Vb^T: S [r], Ubt [r, r] (row i = i-th left vector), SH = diag(S) Vb^T [r, p]
        double *Sg = rowsum, *Ubt = Vt;    // Vt is free once Vtr exists
        CP_HIP(ctx, hipMemsetAsync(SH, 0, g_c * 8, ctx->stream));
        CP_TRY(cp_svd_rows_impl(ctx, B, p_pad, r, p_pad, r, Sg, Ubt, r_pad, SH, p_pad, sc, nullptr));
        std::vector<double> hs(r);
        CP_TRY(read_back(ctx, hs.data(), Sg, size_t(r) * 8));
        const double cut = double(std::max<int64_t>(N, p)) * eps * hs[0];     // _base.py:700-702 / gelsd's rcond
        while (k < r && hs[k] > cut) ++k;
        if (k > 0) {
            const int k_pad = int(cp_align_up(size_t(k), 128));
            double *Ubtr = Gw;   // [r_pad, k_pad]: Ubtr[j, i] = Ubt[i, j], i < k   (Vtr no longer needed)
            k_transpose_pad<<<dim3(k_pad / 32, r_pad / 32), RT, 0, ctx->stream>>>(Ubt, k, r, r_pad, Ubtr, k_pad);
            CP_LAUNCH_CHECK(ctx);
            CP_TRY(cp_gemm_tn_f64(ctx, k_pad, n_pad, r_pad, 1.0, Ubtr, k_pad, Cm, n_pad, 0.0, Dm, n_pad, CP_TRI_NONE));
            k_scale_rows<<<k_pad, RT, 0, ctx->stream>>>(Dm, n_pad, k, n_pad, Sg, 2);   // / S_i^2: SH rows carry one S_i
            CP_LAUNCH_CHECK(ctx);
            const int k16 = int(cp_align_up(size_t(k), 16));
            CP_TRY(cp_gemm_tn_f64(ctx, p_pad, n_pad, k16, 1.0, SH, p_pad, Dm, n_pad, 0.0, rs.Rm, n_pad, CP_TRI_NONE));
        }
    }
    if (k == 0) CP_HIP(ctx, hipMemsetAsync(rs.Rm, 0, r_c * 8, ctx->stream));
    cp_stage_mark(ctx, "refit_robust_jacobi");
    CP_HIP(ctx, hipMemsetAsync(rs.dinfo, 0, sizeof(int), ctx->stream));
    CP_TRY(finalize());
    info->p = p;
    info->rank = k;
    info->fallback = 3;
    info->reserved = 0;
    return CP_OK;
}

template <class NormalEquations>
int refit_solve_tail(cp_ctx *ctx, const RefitSolve &rs, NormalEquations &&normal_equations, cp_refit_info *info) {
    double *const G = rs.G, *const G0 = rs.G0, *const Lt = rs.Lt, *const Uf = rs.Uf, *const Yt = rs.Yt, *const Rm = rs.Rm,
                 *const R2 = rs.R2, *const TI = rs.TI, *const TIT = rs.TIT, *const xmean = rs.xmean, *const ymean = rs.ymean,
                 *const dg0 = rs.dg0, *const gmax = rs.gmax, *const W_out = rs.W_out, *const b_out = rs.b_out,
                 *const b_host = rs.b_host, *const W_host = rs.W_host;
    int *const dinfo = rs.dinfo, *const info_host = rs.info_host;
    const int p = rs.p, p_pad = rs.p_pad, n = rs.n, n_pad = rs.n_pad, nblk = rs.nblk;
    const int64_t N = rs.N;
    const double ridge = rs.ridge;
    const size_t g_b = size_t(p_pad) * p_pad * 8, r_b = size_t(p_pad) * n_pad * 8;
    Chol ch{G, Uf, Lt, TI, TIT, dg0, gmax, dinfo, p, p_pad, nblk};
    int hinfo = 0;
    bool fallback = (ridge == 0.0) && (N - 1 < p);  // centred X has rank <= N-1
    auto finalize = [&]() -> int {
        CP_TRY(finalize_launch(ctx, Rm, n_pad, p, n, xmean, ymean, W_out, b_out, W_host, b_host, dinfo, info_host, Yt));
        cp_stage_mark(ctx, "refit_finalize");
        CP_HIP(ctx, cp_stream_wait(ctx));  // the only wait of the call; everything small came back with the kernel
        hinfo = *info_host;
        return CP_OK;
    };
    if (!fallback) {
        CP_TRY(normal_equations(G, Rm, true));
        if (ctx->defer_refit_wait) {  // cp_prune_layers factors and substitutes all its layers with one launch each
            ctx->deferred = cp_refit_deferred{G, Uf, Lt, TI, TIT, dg0, gmax, Rm, Yt, dinfo, p, p_pad, nblk, n, n_pad, xmean, ymean,
                                              W_out, b_out, W_host, b_host, info_host};
            ctx->refit_pending = true;
            info->p = p;
            info->rank = p;
            info->fallback = 0;
            info->reserved = 0;
            return CP_OK;
        }
        bool fwd = false;   // the forward substitution rode in the launches of the factorisation
        cp_stage_mark(ctx, "refit_chol_begin");   // opens the bracket of the factorisation chain (timing mode 2)
        CP_TRY(chol_factor(ctx, ch, PIV_TOL, Rm, n_pad, &fwd));
        cp_stage_mark(ctx, "refit_cholesky");
        if (nblk >= solve_blocked_min_blocks()) {   // large factor: banded substitution with GEMM updates; the lay-out rides along
            StripFinal fin{p, n, xmean, ymean, W_out, b_out, W_host, b_host, dinfo, info_host};
            bool fin_done = false;
            CP_TRY(chol_solve_blocked(ctx, ch, Rm, n_pad, fwd ? 2 : 3, &fin, &fin_done));
            cp_stage_mark(ctx, "refit_solve");
            if (fin_done) {
                CP_HIP(ctx, cp_stream_wait(ctx));  // the only wait of the call; everything came back with the kernels
                hinfo = *info_host;
            } else {
                CP_TRY(finalize());
            }
        } else {
            StripFinal fin{p, n, xmean, ymean, W_out, b_out, W_host, b_host, dinfo, info_host};
            CP_TRY(chol_solve(ctx, ch, Rm, Yt, n_pad, fin, fwd ? 2 : 3));  // substitution(s) + coefficient lay-out + intercept in one launch
            cp_stage_mark(ctx, "refit_solve");
            CP_HIP(ctx, cp_stream_wait(ctx));  // the only wait of the call; everything small came back with the kernel
            hinfo = *info_host;                // outputs are overwritten below if a pivot failed
        }
        if (hinfo != 0) fallback = true;
    }
    int rank = p;
    if (fallback && ridge == 0.0 && rs.Xs != nullptr) {
        CP_TRY(refit_robust(ctx, rs, normal_equations, finalize, info));
        if (hinfo != 0) return cp_set_error(ctx, CP_ERR_NUMERIC, "refit: rank-revealing path failed (%d)", hinfo);
        return CP_OK;
    }
    if (fallback) {
        // iterated Tikhonov on an untouched Gram G0 and right-hand side R2 (recomputed: this path is rare)
        double *Wacc = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
        if (!Wacc) return cp_set_error(ctx, CP_ERR_NOMEM, "refit: arena (fallback)");
        CP_TRY(normal_equations(G0, R2, false));
        CP_HIP(ctx, hipMemcpyAsync(G, G0, g_b, hipMemcpyDeviceToDevice, ctx->stream));
        k_add_diag_scaled<<<(p + RT - 1) / RT, RT, 0, ctx->stream>>>(G, p_pad, p, gmax, 1e-9, dg0, dinfo, chol_info_count(nblk));
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(chol_factor(ctx, ch, 0.0));
        CP_HIP(ctx, hipMemsetAsync(Wacc, 0, r_b, ctx->stream));
        const size_t cnt = size_t(p_pad) * n_pad;
        const int ab = int(std::min<size_t>((cnt + RT - 1) / RT, size_t(ctx->cu_count) * 8));
        for (int sweep = 0; sweep < 5; ++sweep) {
            CP_HIP(ctx, hipMemcpyAsync(Rm, R2, r_b, hipMemcpyDeviceToDevice, ctx->stream));
            if (sweep > 0)  // Rm = R - G0 W   (G0 symmetric: G0^T W)
                CP_TRY(cp_gemm_tn_f64(ctx, p_pad, n_pad, p_pad, -1.0, G0, p_pad, Wacc, n_pad, 1.0, Rm, n_pad,
                                      CP_TRI_NONE));
            CP_TRY(chol_solve(ctx, ch, Rm, Yt, n_pad));
            k_axpy<<<ab, RT, 0, ctx->stream>>>(Wacc, Rm, cnt);
            CP_LAUNCH_CHECK(ctx);
        }
        CP_HIP(ctx, hipMemcpyAsync(Rm, Wacc, r_b, hipMemcpyDeviceToDevice, ctx->stream));
        cp_stage_mark(ctx, "refit_minnorm_fallback");
        CP_TRY(finalize());
        if (hinfo != 0)
            return cp_set_error(ctx, CP_ERR_NUMERIC, "refit: regularised factorisation broke down at column %d",
                                hinfo - 1);
        rank = -1;  // not determined on this path
    }
    info->p = p;
    info->rank = rank;
    info->fallback = fallback ? 1 : 0;
    info->reserved = 0;
    return CP_OK;
}

}  // namespace

int cp_refit_batch_factor_solve(cp_ctx *const *ctxs, int n_ctx) {
    if (!ctxs || n_ctx <= 0 || n_ctx > CP_REFIT_MAX_BATCH) return CP_ERR_ARG;
    StripBatch sb;
    memset(&sb, 0, sizeof(sb));
    int nj = 0, max_strips = 0;
    cp_ctx *ctx0 = nullptr;
    bool fwd_all = true;   // every layer's forward substitution rode in its factorisation's launches
    for (int l = 0; l < n_ctx; ++l) {
        cp_ctx *c = ctxs[l];
        if (!c || !c->refit_pending) continue;
        if (!ctx0) ctx0 = c;
        const cp_refit_deferred &d = c->deferred;
        const bool blocked = d.nblk >= solve_blocked_min_blocks();   // large factor: banded substitution below, not a strip job
        sb.j[nj] = StripJob{d.U, d.Lt, d.p_pad, d.TI, d.TIT, d.nblk, d.Rm, blocked ? 0 : d.n_pad,
                            StripFinal{d.p, d.n, d.xmean, d.ymean, d.W_out, d.b_out, d.W_host, d.b_host, d.info, d.info_host}};
        if (!blocked) max_strips = std::max(max_strips, d.n_pad / 16);
        ++nj;
        Chol ch{d.G, d.U, d.Lt, d.TI, d.TIT, d.dg0, d.gmax, d.info, d.p, d.p_pad, d.nblk};
        bool fwd = false;
        CP_TRY(chol_factor(c, ch, PIV_TOL, d.Rm, d.n_pad, &fwd));
        fwd_all = fwd_all && fwd;
    }
    if (nj == 0) return CP_OK;
    cp_ctx *ctx = ctx0;
    // the (backward) substitutions of every layer of the batch: one launch (no workgroup waits for another one)
    if (max_strips > 0) {
        if (fwd_all)
            k_solve_strips_batch<2><<<dim3(max_strips, nj), 512, 0, ctx->stream>>>(sb);
        else
            k_solve_strips_batch<3><<<dim3(max_strips, nj), 512, 0, ctx->stream>>>(sb);
        CP_LAUNCH_CHECK(ctx);
    }
    for (int l = 0; l < n_ctx; ++l) {   // large factors: banded substitution with GEMM updates + the lay-out kernel, layer by layer
        cp_ctx *c = ctxs[l];
        if (!c || !c->refit_pending) continue;
        const cp_refit_deferred &d = c->deferred;
        if (d.nblk < solve_blocked_min_blocks()) continue;
        Chol ch{d.G, d.U, d.Lt, d.TI, d.TIT, d.dg0, d.gmax, d.info, d.p, d.p_pad, d.nblk};
        CP_TRY(chol_solve_blocked(c, ch, d.Rm, d.n_pad, fwd_all ? 2 : 3));
        CP_TRY(finalize_launch(c, d.Rm, d.n_pad, d.p, d.n, d.xmean, d.ymean, d.W_out, d.b_out, d.W_host, d.b_host, d.info,
                               d.info_host, d.part));
    }
    return CP_OK;
}

// ---- full normal equations during the alpha search ------------------------------------------------------------
namespace {

// The kept-channel list from the mask, handed over as a KERNEL ARGUMENT (a bit per channel, c <= 2048 -> 256 B of kernarg):
// chan[j] = index of the j-th kept channel.  Replaces a hipMemcpyAsync out of pageable host memory at the one point of a
// layer where everything waits for the host -- between the end of the alpha search and the first launch of the refit: the
// runtime stages such a copy through its own page-locked buffer and a copy kernel (rocprof timeline of the vgg16 job:
// 0.4-1.3 ms between the end of the search kernel and that copy kernel, the copy kernel itself 120-550 us on the busy chip).
__global__ void __launch_bounds__(256) k_chan_from_bits(ChanBits bits, int c, int *__restrict__ chan) {
    chan_from_bits(bits, c, chan);
}
ChanBits chan_bits(const std::vector<int> &chan) {
    ChanBits bits;
    memset(&bits, 0, sizeof(bits));
    for (int ch : chan) bits.w[ch >> 6] |= 1ull << (ch & 63);
    return bits;
}
// the kept-channel list on the device, ordered on ctx->stream
int upload_chan(cp_ctx *ctx, const std::vector<int> &chan, int c, int *dchan) {
    if (c <= CHAN_BITS_MAX) {
        k_chan_from_bits<<<1, 256, 0, ctx->stream>>>(chan_bits(chan), c, dchan);
        CP_LAUNCH_CHECK(ctx);
        return CP_OK;
    }
    CP_HIP(ctx, hipMemcpyAsync(dchan, chan.data(), chan.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return CP_OK;
}

__global__ void __launch_bounds__(RT) k_iota(int *__restrict__ v, int count) {
    const int i = blockIdx.x * RT + threadIdx.x;
    if (i < count) v[i] = i;
}
// G[i, j] = Gf[col(i), col(j)], R[i, :] = Rf[col(i), :], xmean[i] = xf[col(i)] for i, j < p (col(i) = chan[i / kk] kk + i % kk);
// zero in the padding.  One workgroup per output row i < p_pad.  BITS: the channel list arrives as a bit mask in the kernel
// arguments and is unpacked into LDS (no list upload in front of this launch).
template <bool BITS>
__global__ void __launch_bounds__(RT) k_gather_normal_eq(const double *__restrict__ Gf, int ldf, const double *__restrict__ Rf,
                                                         const double *__restrict__ xf, const int *__restrict__ chan_list,
                                                         ChanBits bits, int c, int kk, int p, int p_pad, int n_pad,
                                                         double *__restrict__ G, double *__restrict__ R,
                                                         double *__restrict__ xmean) {
    __shared__ int chan_lds[BITS ? CHAN_BITS_MAX : 1];
    const int *chan = chan_list;
    if (BITS) {
        chan_from_bits(bits, c, chan_lds);
        __syncthreads();
        chan = chan_lds;
    }
    const int i = blockIdx.x;
    const bool live = i < p;
    const int ci = live ? chan[i / kk] * kk + i % kk : 0;
    for (int j = threadIdx.x; j < p_pad; j += RT) {
        double v = 0.0;
        if (live && j < p) v = Gf[size_t(ci) * ldf + chan[j / kk] * kk + j % kk];
        G[size_t(i) * p_pad + j] = v;
    }
    for (int t = threadIdx.x; t < n_pad; t += RT) R[size_t(i) * n_pad + t] = live ? Rf[size_t(ci) * n_pad + t] : 0.0;
    if (threadIdx.x == 0) xmean[i] = live ? xf[ci] : 0.0;
}

}  // namespace

void cp_precompute_void(cp_ctx *ctx) {
    cp_precompute &pc = ctx->pre;
    if (pc.ready || pc.factored) {
        if (pc.worker) hipStreamSynchronize(pc.worker->stream);
        if (pc.chain_stream) hipStreamSynchronize(pc.chain_stream);
    }
    pc.ready = false;
    pc.factored = false;
}

void cp_precompute_release(cp_ctx *ctx) {
    cp_precompute &pc = ctx->pre;
    if (pc.worker) {
        hipStreamSynchronize(pc.worker->stream);
        if (pc.worker->arena) hipFree(pc.worker->arena);
        if (pc.worker->pinned) hipHostFree(pc.worker->pinned);
        // everything else a worker context can have acquired lazily (the split GEMM's arrival counters, ...)
        if (pc.worker->gemm_cnt) hipFree(pc.worker->gemm_cnt);
        if (pc.worker->layer_ws) hipFree(pc.worker->layer_ws);
        if (pc.worker->cd_box) hipFree(pc.worker->cd_box);
        if (pc.worker->stage) hipHostFree(pc.worker->stage);
        if (pc.worker->ev_upload) hipEventDestroy(pc.worker->ev_upload);
        if (pc.worker->ev_fork) hipEventDestroy(pc.worker->ev_fork);
        if (pc.worker->ev_join) hipEventDestroy(pc.worker->ev_join);
        for (int i = 0; i < 2 * CP_MAX_STAGES; ++i)
            if (pc.worker->ev[i]) hipEventDestroy(pc.worker->ev[i]);
        delete pc.worker;
        pc.worker = nullptr;
    }
    if (pc.chain_stream) {
        hipStreamSynchronize(pc.chain_stream);
        hipStreamDestroy(pc.chain_stream);
    }
    if (pc.buf) hipFree(pc.buf);
    if (pc.fbuf) hipFree(pc.fbuf);
    if (pc.done) hipEventDestroy(pc.done);
    if (pc.gram_done) hipEventDestroy(pc.gram_done);
    pc = cp_precompute{};
}

namespace {
// both substitutions (or one of them: sweeps bit 0 forward, bit 1 backward) with the form that suits the factor's size
int chol_solve_any(cp_ctx *ctx, const Chol &ch, double *Rm, int n_pad, int sweeps) {
    if (ch.nblk >= solve_blocked_min_blocks()) return chol_solve_blocked(ctx, ch, Rm, n_pad, sweeps);
    return chol_solve(ctx, ch, Rm, nullptr, n_pad, StripFinal{}, sweeps);
}
bool prefactor_wanted() {
    static const bool on = !(getenv("CP_REFIT_PREFACTOR") && getenv("CP_REFIT_PREFACTOR")[0] == '0');
    return on;
}
// CP_REFIT_FUSED_XTY=0 (or cp_debug_knob(CP_KNOB_SPLIT_XTY, 1)): Gram and X^T Y as two launches, as before round 6
bool fused_xty_wanted() {
    static const bool on = !(getenv("CP_REFIT_FUSED_XTY") && getenv("CP_REFIT_FUSED_XTY")[0] == '0');
    return on && cp_knob(CP_KNOB_SPLIT_XTY) == 0;
}
}  // namespace

// Enqueue, on the device's shared side stream, the normal equations of the layer over ALL c channels:
//   xmean_all, ymean, G_full = Xc^T Xc (P x P, P = c kk), R_full = Xc^T Yc (P x n)  -- 1 / (kept fraction)^2 times the
// flops of the masked Gram, but off the critical path: the alpha search that decides the mask is one workgroup busy for
// milliseconds, the rest of the chip is idle meanwhile.  Ordered after everything already on ctx->stream; the refit
// (cp_lstsq_refit_impl) waits for it and gathers.  Skipped (returns CP_OK, pre.ready = false) when N - 1 < P.
int cp_refit_precompute_enqueue(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const double *Y, int n,
                                double rank_hint, bool fork_recorded) {
    cp_precompute &pc = ctx->pre;
    pc.ready = false;
    pc.factored = false;
    const int P = c * kk;
    if (N - 1 < P) return CP_OK;
    const int P_pad = int(cp_align_up(size_t(P), NB)), n_pad = int(cp_align_up(size_t(n), 128));
    const int64_t N_pad = int64_t(cp_align_up(size_t(N), 16));
    if (!pc.worker) {
        cp_ctx *w = new cp_ctx();
        w->device = ctx->device;
        w->cu_count = ctx->cu_count;
        w->own_stream = nullptr;
        w->stream = cp_side_stream(ctx);
        for (int i = 0; i < 2 * CP_MAX_STAGES; ++i) hipEventCreate(&w->ev[i]);
        w->timing = ctx->timing;
        w->timing_gram_only = ctx->timing_gram_only;
        pc.worker = w;
        CP_HIP(ctx, hipEventCreateWithFlags(&pc.done, hipEventDisableTiming));
    }
    if (!ctx->ev_fork) {
        CP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        CP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    cp_ctx *w = pc.worker;
    const size_t elems = size_t(P_pad) + n_pad + size_t(P_pad) * P_pad + size_t(P_pad) * n_pad;
    if (elems * 8 > pc.buf_bytes) {
        CP_HIP(ctx, hipStreamSynchronize(w->stream));
        CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (pc.buf) CP_HIP(ctx, hipFree(pc.buf));
        pc.buf = nullptr;
        pc.buf_bytes = 0;
        if (hipMalloc(reinterpret_cast<void **>(&pc.buf), elems * 8) != hipSuccess)
            return cp_set_error(ctx, CP_ERR_NOMEM, "refit precompute: hipMalloc(%zu)", elems * 8);
        pc.buf_bytes = elems * 8;
    }
    pc.xmean = reinterpret_cast<double *>(pc.buf);
    pc.ymean = pc.xmean + P_pad;
    pc.G = pc.ymean + n_pad;
    pc.R = pc.G + size_t(P_pad) * P_pad;
    // worker scratch: centred copies of ALL columns and of Y, partial sums, identity channel list, split-K planes
    const int RB = 64, rows_per_block = int((N + RB - 1) / RB);
    size_t ws = std::max(cp_gemm_tn_workspace(w, P_pad, P_pad, int(N_pad), CP_TRI_LOWER_MIRROR),
                         cp_gemm_tn_workspace(w, P_pad, n_pad, int(N_pad), CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_gram_xty_workspace(w, P_pad, n_pad, int(N_pad)));
    // Few channels are dropped (rank hint >= 0.8 c): also factor the FULL Gram and forward-substitute the right-hand side
    // here; the refit then needs no factorisation of its own after the search (refit_from_full_factor).
    const bool want_factor = prefactor_wanted() && rank_hint >= 0.8 * double(c) && rank_hint < double(c);
    const int nblkF = P_pad / NB;
    if (want_factor) {
        ws = std::max(ws, cp_gemm_tn_workspace(w, P_pad, P_pad, NB, CP_TRI_UPPER));
        ws = std::max(ws, chol_solve_blocked_workspace(w, P_pad, n_pad));
        const size_t g_c = size_t(P_pad) * P_pad, ti_c = size_t(nblkF) * NB * NB;
        const size_t felems = 3 * g_c + 2 * ti_c + size_t(P_pad) * n_pad + size_t(P_pad) + 8;
        const size_t fbytes = felems * 8 + size_t(chol_info_count(nblkF) + 16) * 4;
        if (fbytes > pc.fbuf_bytes) {
            CP_HIP(ctx, hipStreamSynchronize(w->stream));
            if (pc.chain_stream) CP_HIP(ctx, hipStreamSynchronize(pc.chain_stream));
            CP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (pc.fbuf) CP_HIP(ctx, hipFree(pc.fbuf));
            pc.fbuf = nullptr;
            pc.fbuf_bytes = 0;
            if (hipMalloc(reinterpret_cast<void **>(&pc.fbuf), fbytes) != hipSuccess)
                return cp_set_error(ctx, CP_ERR_NOMEM, "refit precompute: hipMalloc(%zu)", fbytes);
            pc.fbuf_bytes = fbytes;
        }
        pc.Gw = reinterpret_cast<double *>(pc.fbuf);
        pc.U = pc.Gw + g_c;
        pc.Lt = pc.U + g_c;
        pc.TI = pc.Lt + g_c;
        pc.TIT = pc.TI + ti_c;
        pc.F = pc.TIT + ti_c;
        pc.dg0 = pc.F + size_t(P_pad) * n_pad;
        pc.gmax = pc.dg0 + P_pad;
        pc.finfo = reinterpret_cast<int *>(pc.gmax + 8);
        pc.nblk = nblkF;
        if (!pc.chain_stream) {
            CP_HIP(ctx, hipStreamCreateWithFlags(&pc.chain_stream, hipStreamNonBlocking));
            CP_HIP(ctx, hipEventCreateWithFlags(&pc.gram_done, hipEventDisableTiming));
        }
    }
    // after whatever still reads the previous precompute; fork_recorded: the caller recorded ev_fork at the point of its own
    // stream the worker has to wait for (cp_prune_layer_h2d: before the alpha search it has already enqueued)
    if (!fork_recorded) CP_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    CP_HIP(ctx, hipStreamWaitEvent(w->stream, ctx->ev_fork, 0));
    if (cp_arena_reserve(w, (size_t(N_pad) * (P_pad + n_pad) + size_t(RB) * (P_pad + n_pad)) * 8 + size_t(c) * 4 + ws + (1 << 16)) != CP_OK)
        return cp_set_error(ctx, CP_ERR_NOMEM, "refit precompute: arena");
    double *Xs = cp_arena_take_t<double>(w, size_t(N_pad) * P_pad), *Yc = cp_arena_take_t<double>(w, size_t(N_pad) * n_pad);
    double *part_x = cp_arena_take_t<double>(w, size_t(RB) * P_pad), *part_y = cp_arena_take_t<double>(w, size_t(RB) * n_pad);
    int *dchan = cp_arena_take_t<int>(w, c);
    if (!Xs || !Yc || !part_x || !part_y || !dchan) return cp_set_error(ctx, CP_ERR_NOMEM, "refit precompute: arena carve");
    cp_stage_begin(w);
    k_iota<<<(c + RT - 1) / RT, RT, 0, w->stream>>>(dchan, c);
    CP_LAUNCH_CHECK(ctx);
    const int gx = (P + RT - 1) / RT, gy = (n + RT - 1) / RT;
    if (x_dtype == CP_F32) {
        k_colsum_xy<float><<<dim3(gx + gy, RB), RT, 0, w->stream>>>(static_cast<const float *>(X), Y, N, c, kk, n, dchan, P, gx,
                                                                    rows_per_block, part_x, P_pad, part_y, n_pad);
    } else {
        k_colsum_xy<double><<<dim3(gx + gy, RB), RT, 0, w->stream>>>(static_cast<const double *>(X), Y, N, c, kk, n, dchan, P, gx,
                                                                     rows_per_block, part_x, P_pad, part_y, n_pad);
    }
    CP_LAUNCH_CHECK(ctx);
    k_mean_finish_xy<<<gx + gy, RT, 0, w->stream>>>(part_x, P_pad, P, part_y, n_pad, n, gx, RB, 1.0 / double(N), pc.xmean, pc.ymean);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(w, "refit_means");
    if (x_dtype == CP_F32)
        k_gather_center_xy<float><<<unsigned(N_pad), RT, 0, w->stream>>>(static_cast<const float *>(X), Y, N, c, kk, n, dchan, P,
                                                                           P_pad, n_pad, pc.xmean, pc.ymean, Xs, Yc);
    else
        k_gather_center_xy<double><<<unsigned(N_pad), RT, 0, w->stream>>>(static_cast<const double *>(X), Y, N, c, kk, n, dchan, P,
                                                                            P_pad, n_pad, pc.xmean, pc.ymean, Xs, Yc);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(w, "refit_gather_center");
    cp_stage_mark(w, "refit_gram_begin");
    bool fused = false;
    if (fused_xty_wanted()) {
        w->gemm_mark = "refit_gram_gemm";
        if (cp_gemm_gram_xty(w, P_pad, n_pad, int(N_pad), Xs, P_pad, Yc, n_pad, pc.G, P_pad, pc.R, n_pad, &fused) != CP_OK)
            return cp_set_error(ctx, CP_ERR_HIP, "refit precompute: %s", w->err);
        if (!fused) w->gemm_mark = nullptr;
    }
    if (!fused) {
        w->gemm_tag = CP_GEMM_REFIT_GRAM;
        w->gemm_mark = "refit_gram_gemm";
        if (cp_gemm_tn_f64(w, P_pad, P_pad, int(N_pad), 1.0, Xs, P_pad, Xs, P_pad, 0.0, pc.G, P_pad, CP_TRI_LOWER_MIRROR) != CP_OK)
            return cp_set_error(ctx, CP_ERR_HIP, "refit precompute: %s", w->err);
        cp_stage_mark(w, "refit_gram_reduce");
        w->gemm_tag = CP_GEMM_REFIT_XTY;
        w->gemm_mark = "refit_xty_gemm";
        if (cp_gemm_tn_f64(w, P_pad, n_pad, int(N_pad), 1.0, Xs, P_pad, Yc, n_pad, 0.0, pc.R, n_pad, CP_TRI_NONE) != CP_OK)
            return cp_set_error(ctx, CP_ERR_HIP, "refit precompute: %s", w->err);
        cp_stage_mark(w, "refit_xty_reduce");
    }
    if (want_factor) {
        // The factorisation is a chain of short dependent launches: on this context's OWN stream, so that two layers'
        // chains run side by side while their long products take turns on the shared one.
        CP_HIP(ctx, hipEventRecord(pc.gram_done, w->stream));
        CP_HIP(ctx, hipStreamWaitEvent(pc.chain_stream, pc.gram_done, 0));
        hipStream_t shared = w->stream;
        w->stream = pc.chain_stream;
        const size_t g_c = size_t(P_pad) * P_pad;
        int rc = CP_OK;
        do {
            if (hipMemcpyAsync(pc.Gw, pc.G, g_c * 8, hipMemcpyDeviceToDevice, w->stream) != hipSuccess ||
                hipMemcpyAsync(pc.F, pc.R, size_t(P_pad) * n_pad * 8, hipMemcpyDeviceToDevice, w->stream) != hipSuccess) {
                rc = CP_ERR_HIP;
                break;
            }
            k_diag_prepare<<<1, 256, 0, w->stream>>>(pc.Gw, P_pad, P, P_pad, 0.0, pc.dg0, pc.gmax, pc.finfo, chol_info_count(nblkF));
            Chol chF{pc.Gw, pc.U, pc.Lt, pc.TI, pc.TIT, pc.dg0, pc.gmax, pc.finfo, P, P_pad, nblkF};
            bool fwd = false;   // F = L^-1 R rides in the launches of the factorisation
            if ((rc = chol_factor(w, chF, PIV_TOL, pc.F, n_pad, &fwd)) != CP_OK) break;
            cp_stage_mark(w, "prefactor_cholesky");
            if (!fwd && (rc = chol_solve_any(w, chF, pc.F, n_pad, 1)) != CP_OK) break;
            cp_stage_mark(w, "prefactor_forward");
        } while (false);
        if (rc == CP_OK && hipEventRecord(pc.done, w->stream) != hipSuccess) rc = CP_ERR_HIP;
        w->stream = shared;
        if (rc != CP_OK) return cp_set_error(ctx, rc, "refit precompute (factor): %s", w->err);
        pc.factored = true;
    } else {
        CP_HIP(ctx, hipEventRecord(pc.done, w->stream));
    }
    pc.X = X; pc.Y = Y; pc.N = N; pc.c = c; pc.kk = kk; pc.n = n; pc.x_dtype = x_dtype;
    pc.P = P; pc.P_pad = P_pad; pc.n_pad = n_pad;
    pc.ready = true;
    return CP_OK;
}

namespace {

// T[col(drop_i) , i] = 1 for i < d (T zeroed before): the columns of the identity that belong to the dropped channels
__global__ void __launch_bounds__(RT) k_unit_columns(double *__restrict__ T, int ldt, const int *__restrict__ drop, int kk, int d) {
    const int i = blockIdx.x * RT + threadIdx.x;
    if (i < d) T[size_t(drop[i / kk] * kk + i % kk) * ldt + i] = 1.0;
}
// Rm[i, :] = Wf[col(i), :], xmean[i] = xf[col(i)] for i < p (col(i) = chan[i / kk] kk + i % kk); zero in the padding
__global__ void __launch_bounds__(RT) k_gather_rows(const double *__restrict__ Wf, const double *__restrict__ xf,
                                                    const int *__restrict__ chan, int kk, int p, int n_pad,
                                                    double *__restrict__ Rm, double *__restrict__ xmean) {
    const int i = blockIdx.x;
    const bool live = i < p;
    const int ci = live ? chan[i / kk] * kk + i % kk : 0;
    for (int t = threadIdx.x; t < n_pad; t += RT) Rm[size_t(i) * n_pad + t] = live ? Wf[size_t(ci) * n_pad + t] : 0.0;
    if (threadIdx.x == 0) xmean[i] = live ? xf[ci] : 0.0;
}

// The refit when the FULL normal equations were factored during the alpha search (cp_refit_precompute_enqueue with a
// rank hint): G = L L^T over all P = c kk columns, F = L^-1 R.  The least-squares problem on the kept columns K is the full
// one with the dropped coefficients D forced to zero,  min 1/2 w^T G w - w^T R  s.t.  E^T w = 0  (E = the unit columns of D):
//     T = L^-1 E,   S = T^T T (= E^T G^-1 E, d x d),   lambda = S^-1 T^T F,   w = L^-T (F - T lambda),   W = w[K]
// -- one forward substitution with d = |D| kk right-hand sides, a d x d factorisation and one backward substitution after
// the search, instead of the Cholesky chain of the p x p kept sub-matrix (p/128 dependent block steps: 4.3 of the 6.2 ms
// between the end of the search and the result at c = 512).  Same normal equations, same conditioning class (error
// ~ cond(G) eps; cond(G_KK) <= cond(G)); any failed pivot -- of the full factor (e.g. a dead channel the LASSO drops) or
// of S -- leaves *done = false and the caller continues with the kept sub-matrix as before.
int refit_from_full_factor(cp_ctx *ctx, cp_precompute &pc, const std::vector<int> &chan, const uint8_t *mask, int c, int kk,
                           int n, double *W_out, double *b_out, cp_refit_info *info, bool host_out, bool *done) {
    *done = false;
    const int kept = int(chan.size()), p = kept * kk, P_pad = pc.P_pad, n_pad = pc.n_pad;
    std::vector<int> drop;
    for (int i = 0; i < c; ++i)
        if (!mask[i]) drop.push_back(i);
    const int d = int(drop.size()) * kk;
    const int d_pad = int(cp_align_up(size_t(std::max(d, 1)), NB));
    if (d_pad > P_pad / 4 + NB) return CP_OK;   // many channels dropped: factoring the kept sub-matrix is cheaper
    const int p_pad = int(cp_align_up(size_t(p), NB)), nblkS = d_pad / NB;
    size_t ws = std::max(cp_gemm_tn_workspace(ctx, d_pad, d_pad, P_pad, CP_TRI_LOWER_MIRROR),
                         cp_gemm_tn_workspace(ctx, d_pad, n_pad, P_pad, CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, P_pad, n_pad, d_pad, CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, d_pad, d_pad, NB, CP_TRI_UPPER));
    ws = std::max(ws, chol_solve_blocked_workspace(ctx, P_pad, std::max(n_pad, d_pad)));
    ws = std::max(ws, chol_solve_blocked_workspace(ctx, d_pad, n_pad));
    const size_t t_c = size_t(P_pad) * d_pad, s_c = size_t(d_pad) * d_pad, ti_c = size_t(nblkS) * NB * NB,
                 r_c = size_t(p_pad) * n_pad;
    const size_t need = (2 * t_c + 3 * s_c + 2 * ti_c + size_t(d_pad) * n_pad + 2 * r_c + size_t(p_pad) + size_t(d_pad) + 8) * 8 +
                        size_t(kept + int(drop.size()) + chol_info_count(nblkS) + 32) * 4 + ws + (1 << 16);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *T = cp_arena_take_t<double>(ctx, t_c), *TT = cp_arena_take_t<double>(ctx, t_c);
    double *S = cp_arena_take_t<double>(ctx, s_c), *SU = cp_arena_take_t<double>(ctx, s_c), *SLt = cp_arena_take_t<double>(ctx, s_c);
    double *STI = cp_arena_take_t<double>(ctx, ti_c), *STIT = cp_arena_take_t<double>(ctx, ti_c);
    double *Cm = cp_arena_take_t<double>(ctx, size_t(d_pad) * n_pad);
    double *Rm = cp_arena_take_t<double>(ctx, r_c), *part = cp_arena_take_t<double>(ctx, r_c);
    double *xmean = cp_arena_take_t<double>(ctx, p_pad), *sdg0 = cp_arena_take_t<double>(ctx, d_pad);
    double *sgmax = cp_arena_take_t<double>(ctx, 8);
    int *dchan = cp_arena_take_t<int>(ctx, kept), *ddrop = cp_arena_take_t<int>(ctx, std::max<size_t>(drop.size(), 1));
    int *sinfo = cp_arena_take_t<int>(ctx, chol_info_count(nblkS) + 16);
    if (!T || !TT || !S || !SU || !SLt || !STI || !STIT || !Cm || !Rm || !part || !xmean || !sdg0 || !sgmax || !dchan || !ddrop || !sinfo)
        return cp_set_error(ctx, CP_ERR_NOMEM, "refit (full factor): arena");
    const size_t pin_b = 64 + (host_out ? (size_t(n) + size_t(n) * p) * sizeof(double) : 0);
    CP_TRY(cp_pinned_reserve(ctx, pin_b));
    int *info_host = reinterpret_cast<int *>(ctx->pinned);
    int *flag_host = info_host + 4;
    double *b_host = host_out ? reinterpret_cast<double *>(ctx->pinned + 64) : nullptr;
    double *W_host = host_out ? b_host + n : nullptr;
    cp_stage_begin(ctx);
    CP_HIP(ctx, hipStreamWaitEvent(ctx->stream, pc.done, 0));
    CP_HIP(ctx, hipMemcpyAsync(flag_host, pc.finfo, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    CP_TRY(upload_chan(ctx, chan, c, dchan));
    if (!drop.empty()) CP_TRY(upload_chan(ctx, drop, c, ddrop));
    CP_HIP(ctx, cp_stream_wait(ctx));
    cp_stage_mark(ctx, "refit_wait_prefactor");
    if (*flag_host != 0) return CP_OK;          // the full Gram is not safely positive definite
    Chol chF{pc.Gw, pc.U, pc.Lt, pc.TI, pc.TIT, pc.dg0, pc.gmax, pc.finfo, pc.P, P_pad, pc.nblk};
    if (d > 0) {
        CP_HIP(ctx, hipMemsetAsync(T, 0, t_c * 8, ctx->stream));
        k_unit_columns<<<(d + RT - 1) / RT, RT, 0, ctx->stream>>>(T, d_pad, ddrop, kk, d);
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(chol_solve_any(ctx, chF, T, d_pad, 1));                                   // T = L^-1 E
        cp_stage_mark(ctx, "refit_constraint_forward");
        CP_TRY(cp_gemm_tn_f64(ctx, d_pad, d_pad, P_pad, 1.0, T, d_pad, T, d_pad, 0.0, S, d_pad, CP_TRI_LOWER_MIRROR));
        CP_TRY(cp_gemm_tn_f64(ctx, d_pad, n_pad, P_pad, 1.0, T, d_pad, pc.F, n_pad, 0.0, Cm, n_pad, CP_TRI_NONE));
        k_diag_prepare<<<1, 256, 0, ctx->stream>>>(S, d_pad, d, d_pad, 0.0, sdg0, sgmax, sinfo, chol_info_count(nblkS));
        CP_LAUNCH_CHECK(ctx);
        Chol chS{S, SU, SLt, STI, STIT, sdg0, sgmax, sinfo, d, d_pad, nblkS};
        CP_TRY(chol_factor(ctx, chS, PIV_TOL));
        CP_TRY(chol_solve_any(ctx, chS, Cm, n_pad, 3));                                   // lambda = S^-1 T^T F
        k_transpose_pad<<<dim3(P_pad / 32, d_pad / 32), RT, 0, ctx->stream>>>(T, P_pad, d_pad, d_pad, TT, P_pad);
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(cp_gemm_tn_f64(ctx, P_pad, n_pad, d_pad, -1.0, TT, P_pad, Cm, n_pad, 1.0, pc.F, n_pad, CP_TRI_NONE));   // F -= T lambda
        cp_stage_mark(ctx, "refit_constraint_schur");
    } else {
        CP_HIP(ctx, hipMemsetAsync(sinfo, 0, sizeof(int), ctx->stream));
    }
    CP_TRY(chol_solve_any(ctx, chF, pc.F, n_pad, 2));                                     // w = L^-T (F - T lambda)
    cp_stage_mark(ctx, "refit_backward");
    k_gather_rows<<<p_pad, RT, 0, ctx->stream>>>(pc.F, pc.xmean, dchan, kk, p, n_pad, Rm, xmean);
    CP_LAUNCH_CHECK(ctx);
    CP_TRY(finalize_launch(ctx, Rm, n_pad, p, n, xmean, pc.ymean, W_out, b_out, W_host, b_host, sinfo, info_host, part));
    cp_stage_mark(ctx, "refit_finalize");
    CP_HIP(ctx, cp_stream_wait(ctx));
    if (*info_host != 0) return CP_OK;          // S lost a pivot: the kept sub-matrix route decides (and overwrites the outputs)
    info->p = p;
    info->rank = p;
    info->fallback = 0;
    info->reserved = 0;
    *done = true;
    return CP_OK;
}

}  // namespace

// host_out: also leave b (n) and W (n x p) in the context's pinned block at offset 64 (cp_prune_layer)
int cp_lstsq_refit_impl(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const uint8_t *mask,
                        const double *Y, int n, double ridge, double *W_out, double *b_out, cp_refit_info *info,
                        bool host_out) {
    if (!ctx || !X || !mask || !Y || !W_out || !b_out || !info) return CP_ERR_ARG;
    if (N <= 0 || c <= 0 || kk <= 0 || n <= 0 || ridge < 0) return cp_set_error(ctx, CP_ERR_ARG, "refit: bad sizes");
    if (x_dtype != CP_F32 && x_dtype != CP_F64) return cp_set_error(ctx, CP_ERR_ARG, "refit: bad dtype");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int> chan;
    for (int i = 0; i < c; ++i)
        if (mask[i]) chan.push_back(i);
    const int kept = int(chan.size());
    if (kept == 0) return cp_set_error(ctx, CP_ERR_ARG, "refit: empty mask");
    const int p = kept * kk;
    const int p_pad = int(cp_align_up(size_t(p), NB)), n_pad = int(cp_align_up(size_t(n), 128));
    const int64_t N_pad = int64_t(cp_align_up(size_t(N), 16));
    const int nblk = p_pad / NB;
    const int RB = 64;
    const int rows_per_block = int((N + RB - 1) / RB);
    {   // the full Gram was factored during the alpha search: constrained solve, no factorisation of the kept sub-matrix
        cp_precompute &pf = ctx->pre;
        if (pf.armed && pf.ready && pf.factored && !ctx->defer_refit_wait && pf.X == X && pf.Y == Y && pf.N == N && pf.c == c && pf.kk == kk &&
            pf.n == n && pf.x_dtype == x_dtype && pf.n_pad == n_pad && ridge == 0.0) {
            bool done = false;
            pf.factored = false;
            CP_TRY(refit_from_full_factor(ctx, pf, chan, mask, c, kk, n, W_out, b_out, info, host_out, &done));
            if (done) {
                pf.ready = false;
                return CP_OK;
            }
        }
    }

    const size_t xs_b = size_t(N_pad) * p_pad * 8, yc_b = size_t(N_pad) * n_pad * 8, g_b = size_t(p_pad) * p_pad * 8,
                 r_b = size_t(p_pad) * n_pad * 8, ti_b = size_t(nblk) * NB * NB * 8,
                 part_b = size_t(RB) * size_t(p_pad + n_pad) * 8;
    size_t ws = std::max(cp_gemm_tn_workspace(ctx, p_pad, p_pad, int(N_pad), CP_TRI_LOWER_MIRROR),
                         cp_gemm_tn_workspace(ctx, p_pad, n_pad, int(N_pad), CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_gram_xty_workspace(ctx, p_pad, n_pad, int(N_pad)));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, p_pad, n_pad, p_pad, CP_TRI_NONE));
    ws = std::max(ws, chol_solve_blocked_workspace(ctx, p_pad, n_pad));
    const size_t need = xs_b + yc_b + 4 * g_b + 4 * r_b + 2 * ti_b + part_b + size_t(p_pad) * 8 * 3 + size_t(n_pad) * 8 +
                        size_t(kept) * 4 + size_t(chol_info_count(nblk)) * 4 + ws + (1 << 16);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *Xs = cp_arena_take_t<double>(ctx, size_t(N_pad) * p_pad);
    double *Yc = cp_arena_take_t<double>(ctx, size_t(N_pad) * n_pad);
    double *G = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *G0 = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);  // Gram kept intact for the fallback sweeps
    double *Lt = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *Uf = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *Yt = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
    double *Rm = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
    double *R2 = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
    double *TI = cp_arena_take_t<double>(ctx, size_t(nblk) * NB * NB);
    double *TIT = cp_arena_take_t<double>(ctx, size_t(nblk) * NB * NB);
    double *part_x = cp_arena_take_t<double>(ctx, size_t(RB) * p_pad);
    double *part_y = cp_arena_take_t<double>(ctx, size_t(RB) * n_pad);
    double *xmean = cp_arena_take_t<double>(ctx, p_pad);
    double *dg0 = cp_arena_take_t<double>(ctx, p_pad);
    double *gmax = cp_arena_take_t<double>(ctx, 8);
    double *ymean = cp_arena_take_t<double>(ctx, n_pad);
    int *dchan = cp_arena_take_t<int>(ctx, kept);
    int *dinfo = cp_arena_take_t<int>(ctx, chol_info_count(nblk) + 16);
    if (!Xs || !Yc || !G || !G0 || !Lt || !Uf || !Yt || !Rm || !R2 || !TI || !TIT || !part_x || !part_y || !xmean || !dg0 ||
        !gmax || !ymean || !dchan || !dinfo)
        return cp_set_error(ctx, CP_ERR_NOMEM, "refit: arena");

    // pinned host block the last kernel writes: [info | b (n) | W (n x p)] when the caller wants host copies
    const size_t pin_b = 64 + (host_out ? (size_t(n) + size_t(n) * p) * sizeof(double) : 0);
    CP_TRY(cp_pinned_reserve(ctx, pin_b));
    int *info_host = reinterpret_cast<int *>(ctx->pinned);
    double *b_host = host_out ? reinterpret_cast<double *>(ctx->pinned + 64) : nullptr;
    double *W_host = host_out ? b_host + n : nullptr;
    cp_stage_begin(ctx);
    // full normal equations already under way on the side stream (cp_refit_precompute_enqueue): one shot
    cp_precompute &pc = ctx->pre;
    const bool from_pre = pc.armed && pc.ready && pc.X == X && pc.Y == Y && pc.N == N && pc.c == c && pc.kk == kk && pc.n == n &&
                          pc.x_dtype == x_dtype && pc.n_pad == n_pad && ridge == 0.0;
    if (!from_pre) cp_precompute_void(ctx);   // a leftover nobody may consume: its side-stream work still reads its X / Y
    pc.ready = false;
    // the kept-channel list in device memory: only for the kernels that do not take it as a bit mask in their arguments
    bool chan_uploaded = false;
    auto need_chan = [&]() -> int {
        if (!chan_uploaded) CP_TRY(upload_chan(ctx, chan, c, dchan));
        chan_uploaded = true;
        return CP_OK;
    };
    if (from_pre) {
        CP_HIP(ctx, hipStreamWaitEvent(ctx->stream, pc.done, 0));
        ymean = pc.ymean;                  // persistent until the next enqueue on this context, i.e. beyond this call
    } else {   // column means, then gather + centre (three launches)
        CP_TRY(need_chan());
        const int gx = (p + RT - 1) / RT, gy = (n + RT - 1) / RT;
        dim3 gs(gx + gy, RB);
        if (x_dtype == CP_F32)
            k_colsum_xy<float><<<gs, RT, 0, ctx->stream>>>(static_cast<const float *>(X), Y, N, c, kk, n, dchan, p, gx,
                                                           rows_per_block, part_x, p_pad, part_y, n_pad);
        else
            k_colsum_xy<double><<<gs, RT, 0, ctx->stream>>>(static_cast<const double *>(X), Y, N, c, kk, n, dchan, p, gx,
                                                            rows_per_block, part_x, p_pad, part_y, n_pad);
        CP_LAUNCH_CHECK(ctx);
        k_mean_finish_xy<<<gx + gy, RT, 0, ctx->stream>>>(part_x, p_pad, p, part_y, n_pad, n, gx, RB, 1.0 / double(N),
                                                          xmean, ymean);
        CP_LAUNCH_CHECK(ctx);
        cp_stage_mark(ctx, "refit_means");
    }
    // The centred rows of the kept channels, staged once as float64 (k_gather_center_xy: N_pad x p_pad): what both long
    // products read.  (Letting the GEMM loader gather / convert / centre X itself -- no staging copy, 174 MB less HBM
    // traffic per c = 512 layer, bit-identical -- was built and measured in round 3: every operand element is then
    // converted once per tile that uses it with 4-byte loads, Gram 2.20 -> 3.77 ms; removed.)
    bool staged = false;
    auto stage_rows = [&]() -> int {
        if (staged) return CP_OK;
        CP_TRY(need_chan());
        if (x_dtype == CP_F32)
            k_gather_center_xy<float><<<unsigned(N_pad), RT, 0, ctx->stream>>>(
                static_cast<const float *>(X), Y, N, c, kk, n, dchan, p, p_pad, n_pad, xmean, ymean, Xs, Yc);
        else
            k_gather_center_xy<double><<<unsigned(N_pad), RT, 0, ctx->stream>>>(
                static_cast<const double *>(X), Y, N, c, kk, n, dchan, p, p_pad, n_pad, xmean, ymean, Xs, Yc);
        CP_LAUNCH_CHECK(ctx);
        staged = true;
        return CP_OK;
    };
    // Gram and right-hand side into (Gd, Rd), diagonal prepared (ridge, unit pad diagonal, dg0, gmax, info = 0)
    auto normal_equations = [&](double *Gd, double *Rd, bool mark) -> int {
        if (from_pre && mark) {   // the kept rows / columns of the precomputed full normal equations
            ctx->last_xty_fused = false;
            if (c <= CHAN_BITS_MAX) {
                k_gather_normal_eq<true><<<p_pad, RT, 0, ctx->stream>>>(pc.G, pc.P_pad, pc.R, pc.xmean, nullptr, chan_bits(chan), c, kk, p,
                                                                        p_pad, n_pad, Gd, Rd, xmean);
            } else {
                CP_TRY(need_chan());
                k_gather_normal_eq<false><<<p_pad, RT, 0, ctx->stream>>>(pc.G, pc.P_pad, pc.R, pc.xmean, dchan, ChanBits{}, c, kk, p,
                                                                         p_pad, n_pad, Gd, Rd, xmean);
            }
            CP_LAUNCH_CHECK(ctx);
            cp_stage_mark(ctx, "refit_gather_normal_eq");
            k_diag_prepare<<<1, 256, 0, ctx->stream>>>(Gd, p_pad, p, p_pad, ridge, dg0, gmax, dinfo, chol_info_count(nblk));
            CP_LAUNCH_CHECK(ctx);
            return CP_OK;
        }
        CP_TRY(stage_rows());
        if (mark) cp_stage_mark(ctx, "refit_gather_center");
        if (mark) cp_stage_mark(ctx, "refit_gram_begin");   // opens the bracket of the refit Gram GEMM (timing mode 2)
        bool fused = false;
        if (fused_xty_wanted()) {   // Gram and X^T Y as ONE launch (cp_gemm_gram_xty); the bracket "refit_gram_gemm" then spans both
            ctx->gemm_mark = mark ? "refit_gram_gemm" : nullptr;
            CP_TRY(cp_gemm_gram_xty(ctx, p_pad, n_pad, int(N_pad), Xs, p_pad, Yc, n_pad, Gd, p_pad, Rd, n_pad, &fused));
            if (!fused) ctx->gemm_mark = nullptr;
        }
        if (!fused) {
            ctx->gemm_tag = CP_GEMM_REFIT_GRAM;
            ctx->gemm_mark = mark ? "refit_gram_gemm" : nullptr;
            CP_TRY(cp_gemm_tn_f64(ctx, p_pad, p_pad, int(N_pad), 1.0, Xs, p_pad, Xs, p_pad, 0.0, Gd, p_pad, CP_TRI_LOWER_MIRROR));
            if (mark) cp_stage_mark(ctx, "refit_gram_reduce");
            ctx->gemm_tag = CP_GEMM_REFIT_XTY;
            ctx->gemm_mark = mark ? "refit_xty_gemm" : nullptr;
            CP_TRY(cp_gemm_tn_f64(ctx, p_pad, n_pad, int(N_pad), 1.0, Xs, p_pad, Yc, n_pad, 0.0, Rd, n_pad, CP_TRI_NONE));
            if (mark) cp_stage_mark(ctx, "refit_xty_reduce");
        }
        ctx->last_xty_fused = fused;
        k_diag_prepare<<<1, 256, 0, ctx->stream>>>(Gd, p_pad, p, p_pad, ridge, dg0, gmax, dinfo, chol_info_count(nblk));
        CP_LAUNCH_CHECK(ctx);
        return CP_OK;
    };

    RefitSolve rs{G, G0, Lt, Uf, Yt, Rm, R2, TI, TIT, xmean, ymean, dg0, gmax, dinfo, p, p_pad, n, n_pad, nblk, N, ridge,
                  W_out, b_out, info_host, b_host, W_host};
    rs.Xs = Xs;
    rs.Yc = Yc;
    rs.N_pad = N_pad;
    return refit_solve_tail(ctx, rs, normal_equations, info);
}

// ---- nonlinear_fc: ReLU-aware reconstruction (lib/decompose.py:671-685, 51-59) -------------------------------
namespace {

// U = Y, Z = relu(Y) in the padded [Nr, n_pad] lay-out (zero outside N x n)
__global__ void __launch_bounds__(RT) k_nl_init(const double *__restrict__ Y, int64_t N, int n, int n_pad,
                                                double *__restrict__ U, double *__restrict__ Z) {
    const int64_t r = blockIdx.x;
    for (int j = threadIdx.x; j < n_pad; j += RT) {
        const double y = (r < N && j < n) ? Y[size_t(r) * n + j] : 0.0;
        U[size_t(r) * n_pad + j] = y;
        Z[size_t(r) * n_pad + j] = fmax(y, 0.0);
    }
}

// dst[col, r] = src[r, col]: 32 x 32 tiles through LDS
__global__ void __launch_bounds__(RT) k_transpose_2d(const double *__restrict__ src, int ld_src, double *__restrict__ dst,
                                                     int ld_dst) {
    __shared__ double t[32][33];
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int y = ty; y < 32; y += 8) t[y][tx] = src[size_t(r0 + y) * ld_src + c0 + tx];
    __syncthreads();
    for (int y = ty; y < 32; y += 8) dst[size_t(c0 + y) * ld_dst + r0 + tx] = t[tx][y];
}

// column sums of the padded U over row blocks (fixed order), then Uc = U - mean
__global__ void __launch_bounds__(RT) k_colsum_u(const double *__restrict__ U, int64_t N, int n, int n_pad,
                                                 int rows_per_block, double *__restrict__ part) {
    const int col = blockIdx.x * RT + threadIdx.x;
    if (col >= n) return;
    const int64_t r0 = int64_t(blockIdx.y) * rows_per_block, r1 = min(N, r0 + rows_per_block);
    double s = 0;
    for (int64_t r = r0; r < r1; ++r) s += U[size_t(r) * n_pad + col];
    part[size_t(blockIdx.y) * n_pad + col] = s;
}
__global__ void __launch_bounds__(RT) k_center_u(const double *__restrict__ U, const double *__restrict__ part,
                                                 int nparts, int64_t N, int n, int n_pad, double inv_n,
                                                 double *__restrict__ umean, double *__restrict__ Uc) {
    const int64_t r = blockIdx.x;
    for (int j = threadIdx.x; j < n_pad; j += RT) {
        double m = 0.0;
        if (j < n) {
            for (int b = 0; b < nparts; ++b) m += part[size_t(b) * n_pad + j];
            m *= inv_n;
            if (r == 0) umean[j] = m;
        }
        Uc[size_t(r) * n_pad + j] = (r < N && j < n) ? U[size_t(r) * n_pad + j] - m : 0.0;
    }
}

// U <- solve_relu(RU + umean, Z, lambda)  (lib/decompose.py:51-59), zero outside N x n
__global__ void __launch_bounds__(RT) k_solve_relu(const double *__restrict__ RUc, const double *__restrict__ umean,
                                                   const double *__restrict__ Z, double lambda, int64_t N, int n,
                                                   int n_pad, double *__restrict__ U) {
    const int64_t r = blockIdx.x;
    for (int j = threadIdx.x; j < n_pad; j += RT) {
        double u = 0.0;
        if (r < N && j < n) {
            const double RU = RUc[size_t(r) * n_pad + j] + umean[j], z = Z[size_t(r) * n_pad + j];
            const double U0 = fmin(RU, 0.0);
            const double Cost0 = z * z + lambda * ((U0 - RU) * (U0 - RU));
            const double U1 = fmax((lambda * RU + z) / (lambda + 1.0), 0.0);
            const double Cost1 = (U1 - z) * (U1 - z) + lambda * ((U1 - RU) * (U1 - RU));
            u = Cost0 <= Cost1 ? U0 : U1;
        }
        U[size_t(r) * n_pad + j] = u;
    }
}

}  // namespace

extern "C" int cp_nonlinear_fc(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const uint8_t *mask,
                               const double *Y, int n, const int *iters, const double *lambdas, int n_stage,
                               double *W_out, double *b_out, cp_refit_info *info) {
    if (!ctx || !X || !mask || !Y || !iters || !lambdas || !W_out || !b_out || !info) return CP_ERR_ARG;
    if (N <= 0 || c <= 0 || kk <= 0 || n <= 0 || n_stage <= 0) return cp_set_error(ctx, CP_ERR_ARG, "nonlinear_fc: bad sizes");
    if (x_dtype != CP_F32 && x_dtype != CP_F64) return cp_set_error(ctx, CP_ERR_ARG, "nonlinear_fc: bad dtype");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int> chan;
    for (int i = 0; i < c; ++i)
        if (mask[i]) chan.push_back(i);
    const int kept = int(chan.size());
    if (kept == 0) return cp_set_error(ctx, CP_ERR_ARG, "nonlinear_fc: empty mask");
    const int p = kept * kk;
    if (N - 1 < p)
        return cp_set_error(ctx, CP_ERR_UNSUPPORTED, "nonlinear_fc: needs N - 1 >= p (full column rank), N=%lld p=%d",
                            (long long)N, p);
    const int p_pad = int(cp_align_up(size_t(p), NB)), n_pad = int(cp_align_up(size_t(n), 128));
    const int64_t Nr = int64_t(cp_align_up(size_t(N), 128));  // sample rows, also an M dimension here
    const int nblk = p_pad / NB;
    const int RB = 64, rows_per_block = int((N + RB - 1) / RB);
    const size_t xs_c = size_t(Nr) * p_pad, u_c = size_t(Nr) * n_pad, g_c = size_t(p_pad) * p_pad, r_c = size_t(p_pad) * n_pad;
    size_t ws = std::max(cp_gemm_tn_workspace(ctx, p_pad, p_pad, int(Nr), CP_TRI_LOWER_MIRROR),
                         cp_gemm_tn_workspace(ctx, p_pad, n_pad, int(Nr), CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, int(Nr), n_pad, p_pad, CP_TRI_NONE));
    const size_t need = (2 * xs_c + 4 * u_c + 3 * g_c + r_c + 2 * size_t(nblk) * NB * NB + size_t(RB) * (p_pad + n_pad) +
                         3 * size_t(p_pad) + 2 * size_t(n_pad) + 64) * 8 + size_t(kept) * 4 + size_t(chol_info_count(nblk) + 16) * 4 + ws +
                        (1 << 16);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *Xs = cp_arena_take_t<double>(ctx, xs_c), *XsT = cp_arena_take_t<double>(ctx, xs_c);
    double *Ub = cp_arena_take_t<double>(ctx, u_c), *Zb = cp_arena_take_t<double>(ctx, u_c);
    double *Uc = cp_arena_take_t<double>(ctx, u_c), *RU = cp_arena_take_t<double>(ctx, u_c);
    double *G = cp_arena_take_t<double>(ctx, g_c), *Uf = cp_arena_take_t<double>(ctx, g_c), *Lt = cp_arena_take_t<double>(ctx, g_c);
    double *Rm = cp_arena_take_t<double>(ctx, r_c);
    double *TI = cp_arena_take_t<double>(ctx, size_t(nblk) * NB * NB), *TIT = cp_arena_take_t<double>(ctx, size_t(nblk) * NB * NB);
    double *part_x = cp_arena_take_t<double>(ctx, size_t(RB) * p_pad), *part_y = cp_arena_take_t<double>(ctx, size_t(RB) * n_pad);
    double *xmean = cp_arena_take_t<double>(ctx, p_pad), *dg0 = cp_arena_take_t<double>(ctx, p_pad);
    double *gmax = cp_arena_take_t<double>(ctx, 8), *umean = cp_arena_take_t<double>(ctx, n_pad);
    double *ydummy = cp_arena_take_t<double>(ctx, n_pad);
    int *dchan = cp_arena_take_t<int>(ctx, kept), *dinfo = cp_arena_take_t<int>(ctx, chol_info_count(nblk) + 16);
    if (!Xs || !XsT || !Ub || !Zb || !Uc || !RU || !G || !Uf || !Lt || !Rm || !TI || !TIT || !part_x || !part_y || !xmean ||
        !dg0 || !gmax || !umean || !ydummy || !dchan || !dinfo)
        return cp_set_error(ctx, CP_ERR_NOMEM, "nonlinear_fc: arena");
    CP_TRY(cp_pinned_reserve(ctx, 4096));
    int *info_host = reinterpret_cast<int *>(ctx->pinned);

    cp_stage_begin(ctx);
    CP_HIP(ctx, hipMemcpyAsync(dchan, chan.data(), size_t(kept) * 4, hipMemcpyHostToDevice, ctx->stream));
    {   // centred kept columns of X (constant over the iterations), its transpose, Gram, Cholesky -- once
        const int gx = (p + RT - 1) / RT;
        dim3 gs(gx, RB);
        if (x_dtype == CP_F32)
            k_colsum_xy<float><<<gs, RT, 0, ctx->stream>>>(static_cast<const float *>(X), Y, N, c, kk, n, dchan, p, gx,
                                                           rows_per_block, part_x, p_pad, part_y, n_pad);
        else
            k_colsum_xy<double><<<gs, RT, 0, ctx->stream>>>(static_cast<const double *>(X), Y, N, c, kk, n, dchan, p, gx,
                                                            rows_per_block, part_x, p_pad, part_y, n_pad);
        CP_LAUNCH_CHECK(ctx);
        k_mean_finish_xy<<<gx, RT, 0, ctx->stream>>>(part_x, p_pad, p, part_y, n_pad, 0, gx, RB, 1.0 / double(N), xmean,
                                                     ydummy);
        CP_LAUNCH_CHECK(ctx);
        CP_HIP(ctx, hipMemsetAsync(ydummy, 0, size_t(n_pad) * 8, ctx->stream));  // gather below centres "Y" by 0: unused
        if (x_dtype == CP_F32)
            k_gather_center_xy<float><<<unsigned(Nr), RT, 0, ctx->stream>>>(static_cast<const float *>(X), Y, N, c, kk, 0,
                                                                            dchan, p, p_pad, 0, xmean, ydummy, Xs, Uc);
        else
            k_gather_center_xy<double><<<unsigned(Nr), RT, 0, ctx->stream>>>(static_cast<const double *>(X), Y, N, c, kk, 0,
                                                                             dchan, p, p_pad, 0, xmean, ydummy, Xs, Uc);
        CP_LAUNCH_CHECK(ctx);
        k_transpose_2d<<<dim3(unsigned(Nr / 32), p_pad / 32), RT, 0, ctx->stream>>>(Xs, p_pad, XsT, int(Nr));
        CP_LAUNCH_CHECK(ctx);
        CP_TRY(cp_gemm_tn_f64(ctx, p_pad, p_pad, int(Nr), 1.0, Xs, p_pad, Xs, p_pad, 0.0, G, p_pad, CP_TRI_LOWER_MIRROR));
        k_diag_prepare<<<1, 256, 0, ctx->stream>>>(G, p_pad, p, p_pad, 0.0, dg0, gmax, dinfo, chol_info_count(nblk));
        CP_LAUNCH_CHECK(ctx);
    }
    Chol ch{G, Uf, Lt, TI, TIT, dg0, gmax, dinfo, p, p_pad, nblk};
    CP_TRY(chol_factor(ctx, ch, 1e-10));
    k_nl_init<<<unsigned(Nr), RT, 0, ctx->stream>>>(Y, N, n, n_pad, Ub, Zb);
    CP_LAUNCH_CHECK(ctx);
    cp_stage_mark(ctx, "nonlinear_setup");
    const int gy = (n + RT - 1) / RT;
    for (int st = 0; st < n_stage; ++st)
        for (int it = 0; it < iters[st]; ++it) {
            // reg = fc_kernel(X, U): centre U, normal equations with the factor computed above
            k_colsum_u<<<dim3(gy, RB), RT, 0, ctx->stream>>>(Ub, N, n, n_pad, rows_per_block, part_y);
            CP_LAUNCH_CHECK(ctx);
            k_center_u<<<unsigned(Nr), RT, 0, ctx->stream>>>(Ub, part_y, RB, N, n, n_pad, 1.0 / double(N), umean, Uc);
            CP_LAUNCH_CHECK(ctx);
            CP_TRY(cp_gemm_tn_f64(ctx, p_pad, n_pad, int(Nr), 1.0, Xs, p_pad, Uc, n_pad, 0.0, Rm, n_pad, CP_TRI_NONE));
            CP_TRY(chol_solve(ctx, ch, Rm, nullptr, n_pad));
            // RU = reg.predict(X) = Xc W + mean(U); U = solve_relu(RU, Z, lambda)
            CP_TRY(cp_gemm_tn_f64(ctx, int(Nr), n_pad, p_pad, 1.0, XsT, int(Nr), Rm, n_pad, 0.0, RU, n_pad, CP_TRI_NONE));
            k_solve_relu<<<unsigned(Nr), RT, 0, ctx->stream>>>(RU, umean, Zb, lambdas[st], N, n, n_pad, Ub);
            CP_LAUNCH_CHECK(ctx);
        }
    cp_stage_mark(ctx, "nonlinear_iterations");
    // coefficients and intercept of the LAST regression (decompose.py:685): W in Rm, mean(U) of that fit in umean
    k_finalize<<<n, RT, 0, ctx->stream>>>(Rm, n_pad, p, n, xmean, umean, W_out, b_out, nullptr, nullptr, dinfo, info_host);
    CP_LAUNCH_CHECK(ctx);
    CP_HIP(ctx, cp_stream_wait(ctx));
    if (*info_host != 0)
        return cp_set_error(ctx, CP_ERR_NUMERIC, "nonlinear_fc: X^T X is not positive definite at column %d (rank-deficient "
                                                 "inputs are not supported on this path)", *info_host - 1);
    info->p = p;
    info->rank = p;
    info->fallback = 0;
    info->reserved = 0;
    return CP_OK;
}

// ---- ITQ_decompose iterations (channel decomposition, lib/decompose.py:163-246) ----------------------------
namespace {

// G = Y - colmean(Y) (padded [Nr, n_pad]) from the partial column sums; also Z = relu(gt) and the means
__global__ void __launch_bounds__(RT) k_itq_init(const double *__restrict__ Yf, const double *__restrict__ gt,
                                                 const double *__restrict__ part, int nparts, int64_t N, int n,
                                                 int n_pad, double inv_n, double *__restrict__ ymean,
                                                 double *__restrict__ G, double *__restrict__ Z) {
    const int64_t r = blockIdx.x;
    for (int j = threadIdx.x; j < n_pad; j += RT) {
        double m = 0.0;
        if (j < n) {
            for (int b = 0; b < nparts; ++b) m += part[size_t(b) * n_pad + j];
            m *= inv_n;
            if (r == 0) ymean[j] = m;
        }
        const bool live = r < N && j < n;
        G[size_t(r) * n_pad + j] = live ? Yf[size_t(r) * n + j] - m : 0.0;
        Z[size_t(r) * n_pad + j] = live ? fmax(gt[size_t(r) * n + j], 0.0) : 0.0;
    }
}

// column sums of an unpadded [N, n] matrix into part[rb][col] (ld n_pad)
__global__ void __launch_bounds__(RT) k_colsum_plain(const double *__restrict__ Y, int64_t N, int n, int n_pad,
                                                     int rows_per_block, double *__restrict__ part) {
    const int col = blockIdx.x * RT + threadIdx.x;
    if (col >= n) return;
    const int64_t r0 = int64_t(blockIdx.y) * rows_per_block, r1 = min(N, r0 + rows_per_block);
    double s = 0;
    for (int64_t r = r0; r < r1; ++r) s += Y[size_t(r) * n + col];
    part[size_t(blockIdx.y) * n_pad + col] = s;
}

// factors of pinv(PG) = sum_{sigma_k > cond sigma_0} v_k v_k^T / sigma_k:  A[k, :] = v_k / sigma_k, B[k, :] = v_k
__global__ void __launch_bounds__(RT) k_pinv_factors(const double *__restrict__ sigma, const double *__restrict__ Vt, int n,
                                                     int n_pad, double cond, double *__restrict__ A,
                                                     double *__restrict__ B) {
    const int k = blockIdx.x;  // < kp rows (zero beyond n)
    const bool keep = k < n && sigma[k] > cond * sigma[0];
    const double inv = keep ? 1.0 / sigma[k] : 0.0;
    for (int j = threadIdx.x; j < n_pad; j += RT) {
        const double v = (k < n && j < n) ? Vt[size_t(k) * n + j] : 0.0;
        A[size_t(k) * n_pad + j] = v * inv;
        B[size_t(k) * n_pad + j] = keep ? v : 0.0;
    }
}

__global__ void __launch_bounds__(RT) k_pad_rows(const double *__restrict__ src, int rows, int cols, int ld_src,
                                                 double *__restrict__ dst, int ld_dst) {
    const int r = blockIdx.x;
    for (int col = threadIdx.x; col < ld_dst; col += RT)
        dst[size_t(r) * ld_dst + col] = (r < rows && col < cols) ? src[size_t(r) * ld_src + col] : 0.0;
}

}  // namespace

// The 30 + 20 iterations of ITQ_decompose (decompose.py:163-246): feature / gt_feature DEVICE [N, n] f64.
// Outputs DEVICE: T [n, n] (the last low-rank map, decompose.py:226), ymean [n] (Y_mean), umean [n] (the last U_mean).
extern "C" int cp_itq_iterate(cp_ctx *ctx, const double *feature, const double *gt_feature, int64_t N, int n, int rank,
                              const int *iters, const double *lambdas, int n_stage, double pinv_cond, double *T_out,
                              double *ymean_out, double *umean_out) {
    if (!ctx || !feature || !gt_feature || !iters || !lambdas || !T_out || !ymean_out || !umean_out) return CP_ERR_ARG;
    if (N <= 0 || n <= 0 || rank <= 0 || rank > n || n > N || n_stage <= 0)
        return cp_set_error(ctx, CP_ERR_ARG, "itq: bad sizes (needs 0 < rank <= n <= N)");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    const int np_ = int(cp_align_up(size_t(n), 128)), rp = int(cp_align_up(size_t(rank), 16));
    const int kp = int(cp_align_up(size_t(n), 16));
    const int64_t Nr = int64_t(cp_align_up(size_t(N), 128));
    const int RB = 64, rows_per_block = int((N + RB - 1) / RB);
    const size_t big = size_t(Nr) * np_, sq = size_t(np_) * np_;
    size_t ws = cp_gemm_tn_workspace(ctx, np_, np_, int(Nr), CP_TRI_NONE);
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, int(Nr), np_, np_, CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, np_, int(Nr), np_, CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, np_, np_, np_, CP_TRI_NONE));
    ws = std::max(ws, cp_gemm_tn_workspace(ctx, np_, np_, rp, CP_TRI_NONE));
    const size_t need = (7 * big + 16 * sq + size_t(rp) * np_ + size_t(RB) * np_ + 4 * size_t(np_) + 2 * size_t(n) * n +
                         cp_sign_workspace_doubles(np_)) * 8 + SvdScratch::bytes(n, np_) + ws + (1 << 18);
    CP_TRY(cp_arena_reserve(ctx, need));
    double *G = cp_arena_take_t<double>(ctx, big), *GT = cp_arena_take_t<double>(ctx, big);
    double *P1 = cp_arena_take_t<double>(ctx, big), *UU = cp_arena_take_t<double>(ctx, big);
    double *Ub = cp_arena_take_t<double>(ctx, big), *Zb = cp_arena_take_t<double>(ctx, big);
    double *Tn = cp_arena_take_t<double>(ctx, big);                                  // RU [Nr, np_]
    double *GtG = cp_arena_take_t<double>(ctx, sq), *PGi = cp_arena_take_t<double>(ctx, sq);
    double *PiT = cp_arena_take_t<double>(ctx, sq), *T2 = cp_arena_take_t<double>(ctx, sq);
    double *FA = cp_arena_take_t<double>(ctx, sq), *FB = cp_arena_take_t<double>(ctx, sq);
    double *BT = cp_arena_take_t<double>(ctx, sq), *Bm = cp_arena_take_t<double>(ctx, sq);
    double *C1 = cp_arena_take_t<double>(ctx, sq), *Mx = cp_arena_take_t<double>(ctx, sq);
    double *Pr = cp_arena_take_t<double>(ctx, sq), *BP = cp_arena_take_t<double>(ctx, sq);
    double *Rp = cp_arena_take_t<double>(ctx, sq), *RpT = cp_arena_take_t<double>(ctx, sq), *Wp = cp_arena_take_t<double>(ctx, sq);
    double *Vtp = cp_arena_take_t<double>(ctx, size_t(rp) * np_);
    double *part = cp_arena_take_t<double>(ctx, size_t(RB) * np_);
    double *ymean = cp_arena_take_t<double>(ctx, np_), *umean = cp_arena_take_t<double>(ctx, np_);
    double *sigma = cp_arena_take_t<double>(ctx, np_), *Vt = cp_arena_take_t<double>(ctx, size_t(n) * n);
    double *SHsq = cp_arena_take_t<double>(ctx, size_t(n) * np_);
    double *sign_work = cp_arena_take_t<double>(ctx, cp_sign_workspace_doubles(np_));
    SvdScratch sc;
    if (!sign_work || !G || !GT || !P1 || !UU || !Ub || !Zb || !Tn || !GtG || !PGi || !PiT || !T2 || !FA || !FB || !BT || !Bm || !C1 || !Mx ||
        !Pr || !BP || !Rp || !RpT || !Wp || !Vtp || !part || !ymean || !umean || !sigma || !Vt || !SHsq || !sc.take(ctx, n, np_))
        return cp_set_error(ctx, CP_ERR_NOMEM, "itq: arena");
    cp_stage_begin(ctx);
    const int gy = (n + RT - 1) / RT;
    const int me = cp_svd_me(n);
    // G = Y - Y_mean, Z = relu(gt); G^T; GtG = G^T G; PGi = pinv(GtG); P1 = G PGi (= PGGt^T); PiT = GtG PGi
    k_colsum_plain<<<dim3(gy, RB), RT, 0, ctx->stream>>>(feature, N, n, np_, rows_per_block, part);
    CP_LAUNCH_CHECK(ctx);
    k_itq_init<<<unsigned(Nr), RT, 0, ctx->stream>>>(feature, gt_feature, part, RB, N, n, np_, 1.0 / double(N), ymean, G, Zb);
    CP_LAUNCH_CHECK(ctx);
    k_transpose_2d<<<dim3(unsigned(Nr / 32), np_ / 32), RT, 0, ctx->stream>>>(G, np_, GT, int(Nr));
    CP_LAUNCH_CHECK(ctx);
    CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, int(Nr), 1.0, G, np_, G, np_, 0.0, GtG, np_, CP_TRI_NONE));
    int sweeps = 0;
    CP_TRY(cp_svd_rows_impl(ctx, GtG, np_, n, n, n, sigma, Vt, n, SHsq, n, sc, &sweeps));
    CP_HIP(ctx, hipMemsetAsync(FA, 0, sq * 8, ctx->stream));
    CP_HIP(ctx, hipMemsetAsync(FB, 0, sq * 8, ctx->stream));
    k_pinv_factors<<<kp, RT, 0, ctx->stream>>>(sigma, Vt, n, np_, pinv_cond, FA, FB);
    CP_LAUNCH_CHECK(ctx);
    CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, kp, 1.0, FA, np_, FB, np_, 0.0, PGi, np_, CP_TRI_NONE));
    CP_TRY(cp_gemm_tn_f64(ctx, int(Nr), np_, np_, 1.0, GT, int(Nr), PGi, np_, 0.0, P1, np_, CP_TRI_NONE));
    CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, np_, 1.0, GtG, np_, PGi, np_, 0.0, PiT, np_, CP_TRI_NONE));
    // UU = G, U_mean = Y_mean
    CP_HIP(ctx, hipMemcpyAsync(UU, G, big * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CP_HIP(ctx, hipMemcpyAsync(umean, ymean, size_t(np_) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    cp_stage_mark(ctx, "itq_setup");
    // The reference takes the rank-truncated SVD of X = G (PGGt UU) [N, n] (decompose.py:218-220): T_X = L_r S_r R_r =
    // X V_r V_r^T with V_r the leading right singular vectors = leading eigenvectors of X^T X = B^T (G^T G) B, B = PGGt UU
    // [n, n].  Everything the iteration needs from it is n x n:  T = PGGt T_X = (PGi GtG) B V_r V_r^T, so the 5000-row
    // Jacobi SVD per alternation (2.5 s per conv3 layer) becomes an n x n symmetric eigenproblem, warm-started from the
    // previous alternation's rotation (the iterate moves little: 2-3 sweeps instead of 8-10).  Two N-sized products per
    // alternation are left: B^T = UU^T P1 and RU = G T.
    // From the third alternation on the projector V_r V_r^T comes from the matrix sign function instead (sign_ns.hip): the
    // threshold between lambda_r and lambda_{r+1} is carried along as a fraction of the trace, seeded by the eigenvalues of
    // the second alternation's Jacobi run, and every use is verified (trace of the projector = rank); an alternation whose
    // threshold cannot be found goes through the Jacobi sweeps and re-seeds it.  CP_ITQ_SIGN=0: Jacobi throughout.
    // (the first alternation starts from the rotation that diagonalised G^T G for the pseudo-inverse above: UU = G makes
    //  B = pinv(G^T G) G^T G a projector onto eigenvectors of G^T G, so Mx has the same eigenvectors -- 2 sweeps, not 15)
    bool warm = getenv("CP_ITQ_COLD") == nullptr;
    int total_sweeps = 0, alt = 0, sign_alts = 0;
    const double itq_tol = 0.0;   // the standard rounding-level tolerance of the Jacobi sweeps (1e-12 / 1e-10 measured no faster)
    const char *sign_env = getenv("CP_ITQ_SIGN");
    const bool sign_route = !(sign_env && sign_env[0] == '0') && rank < n;
    SignTracker tk;
    for (int st = 0; st < n_stage; ++st)
        for (int it = 0; it < iters[st]; ++it, ++alt) {
            CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, int(Nr), 1.0, UU, np_, P1, np_, 0.0, BT, np_, CP_TRI_NONE));      // B^T
            k_transpose_2d<<<dim3(np_ / 32, np_ / 32), RT, 0, ctx->stream>>>(BT, np_, Bm, np_);                      // B
            CP_LAUNCH_CHECK(ctx);
            CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, np_, 1.0, GtG, np_, Bm, np_, 0.0, C1, np_, CP_TRI_NONE));          // GtG B
            CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, np_, 1.0, Bm, np_, C1, np_, 0.0, Mx, np_, CP_TRI_NONE));           // B^T GtG B
            bool have_projector = false;
            if (sign_route && alt >= 2 && tk.sigma_rel > 0.0) {
                CP_TRY(cp_sign_projector(ctx, Mx, np_, rank, tk, sign_work, Pr, &have_projector));
                if (have_projector) ++sign_alts;
                else warm = false;            // the Jacobi state is from an older alternation: start it from the matrix itself
            }
            if (!have_projector) {
                if (warm) {   // Wk = R_prev Mx, R = R_prev
                    k_pad_rows<<<np_, RT, 0, ctx->stream>>>(sc.R, me, me, me, Rp, np_);
                    CP_LAUNCH_CHECK(ctx);
                    k_transpose_2d<<<dim3(np_ / 32, np_ / 32), RT, 0, ctx->stream>>>(Rp, np_, RpT, np_);
                    CP_LAUNCH_CHECK(ctx);
                    CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, np_, 1.0, RpT, np_, Mx, np_, 0.0, Wp, np_, CP_TRI_NONE));
                    CP_HIP(ctx, hipMemcpy2DAsync(sc.Wk, size_t(np_) * 8, Wp, size_t(np_) * 8, size_t(np_) * 8, size_t(me),
                                                 hipMemcpyDeviceToDevice, ctx->stream));
                }
                CP_TRY(cp_svd_rows_core(ctx, Mx, np_, n, np_, rank, sigma, Vt, n, SHsq, np_, sc, &sweeps, warm, 1e-13, itq_tol));
                warm = getenv("CP_ITQ_COLD") == nullptr;
                total_sweeps += sweeps;
                // threshold for the sign route: the geometric middle of the gap behind lambda_r, as a fraction of the trace
                // (Mx is positive semi-definite: its singular values are its eigenvalues)
                tk.sigma_rel = sc.sigma_sum > 0.0 && sc.sigma_r > 0.0
                                   ? std::sqrt(sc.sigma_r * std::max(sc.sigma_next, 1e-6 * sc.sigma_r)) / sc.sigma_sum : 0.0;
                k_pad_rows<<<rp, RT, 0, ctx->stream>>>(Vt, rank, n, n, Vtp, np_);
                CP_LAUNCH_CHECK(ctx);
                CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, rp, 1.0, Vtp, np_, Vtp, np_, 0.0, Pr, np_, CP_TRI_NONE));       // V_r V_r^T
            }
            CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, np_, 1.0, BT, np_, Pr, np_, 0.0, BP, np_, CP_TRI_NONE));            // B P_r
            CP_TRY(cp_gemm_tn_f64(ctx, np_, np_, np_, 1.0, PiT, np_, BP, np_, 0.0, T2, np_, CP_TRI_NONE));           // T
            // RU = G T + U_mean; U = solve_relu(RU, Z, lambda); U_mean = mean(U); UU = U - U_mean
            CP_TRY(cp_gemm_tn_f64(ctx, int(Nr), np_, np_, 1.0, GT, int(Nr), T2, np_, 0.0, Tn, np_, CP_TRI_NONE));
            k_solve_relu<<<unsigned(Nr), RT, 0, ctx->stream>>>(Tn, umean, Zb, lambdas[st], N, n, np_, Ub);
            CP_LAUNCH_CHECK(ctx);
            k_colsum_u<<<dim3(gy, RB), RT, 0, ctx->stream>>>(Ub, N, n, np_, rows_per_block, part);
            CP_LAUNCH_CHECK(ctx);
            k_center_u<<<unsigned(Nr), RT, 0, ctx->stream>>>(Ub, part, RB, N, n, np_, 1.0 / double(N), umean, UU);
            CP_LAUNCH_CHECK(ctx);
        }
    ctx->itq_sweeps = total_sweeps;
    ctx->itq_ns_steps = int(tk.steps);
    ctx->itq_sign_alternations = sign_alts;
    cp_stage_mark(ctx, "itq_iterations");
    CP_HIP(ctx, hipMemcpy2DAsync(T_out, size_t(n) * 8, T2, size_t(np_) * 8, size_t(n) * 8, size_t(n), hipMemcpyDeviceToDevice,
                                 ctx->stream));
    CP_HIP(ctx, hipMemcpyAsync(ymean_out, ymean, size_t(n) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CP_HIP(ctx, hipMemcpyAsync(umean_out, umean, size_t(n) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CP_HIP(ctx, cp_stream_wait(ctx));
    return CP_OK;
}

extern "C" int cp_lstsq_refit(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const uint8_t *mask,
                              const double *Y, int n, double ridge, double *W_out, double *b_out,
                              cp_refit_info *info) {
    return cp_lstsq_refit_impl(ctx, X, x_dtype, N, c, kk, mask, Y, n, ridge, W_out, b_out, info, false);
}

// ---- sample-sharded refit ("secondary"): rows of X / Y live on several ranks ----------------
// Three calls with one all-reduce between each pair; the caller owns the exchange buffers (cp_refit_shard_layout)
// and runs the collectives (torch.distributed / RCCL) on them:
//   cp_refit_shard_sums   column sums of this rank's rows            -> all-reduce(sum) of sums[p_pad + n_pad]
//   cp_refit_shard_gram   centre with the GLOBAL means, Xs^T Xs and Xs^T Yc of this rank's rows
//                                                                    -> all-reduce(sum) of gram[p_pad (p_pad + n_pad)]
//   cp_refit_shard_solve  the same factor / substitute / lay-out tail as cp_lstsq_refit, on every rank (identical
//                         inputs after the all-reduce => identical outputs, nothing else to exchange)
namespace {

__global__ void __launch_bounds__(RT) k_scale_vec(const double *__restrict__ src, double scale, int count,
                                                  double *__restrict__ dst) {
    const int i = blockIdx.x * RT + threadIdx.x;
    if (i < count) dst[i] = src[i] * scale;
}

struct ShardDims {
    int kept, p, p_pad, n_pad, nblk;
    std::vector<int> chan;
};

int shard_dims(cp_ctx *ctx, int c, int kk, const uint8_t *mask, int n, ShardDims &d) {
    d.chan.clear();
    for (int i = 0; i < c; ++i)
        if (mask[i]) d.chan.push_back(i);
    d.kept = int(d.chan.size());
    if (d.kept == 0) return cp_set_error(ctx, CP_ERR_ARG, "refit: empty mask");
    d.p = d.kept * kk;
    d.p_pad = int(cp_align_up(size_t(d.p), NB));
    d.n_pad = int(cp_align_up(size_t(n), 128));
    d.nblk = d.p_pad / NB;
    return CP_OK;
}

}  // namespace

extern "C" int cp_refit_shard_layout(int kept, int kk, int n, int64_t *sums_elems, int64_t *gram_elems) {
    if (kept <= 0 || kk <= 0 || n <= 0 || !sums_elems || !gram_elems) return CP_ERR_ARG;
    const int64_t p_pad = int64_t(cp_align_up(size_t(kept) * kk, NB)), n_pad = int64_t(cp_align_up(size_t(n), 128));
    *sums_elems = p_pad + n_pad;
    *gram_elems = p_pad * (p_pad + n_pad);
    return CP_OK;
}

extern "C" int cp_refit_shard_sums(cp_ctx *ctx, const void *X, int x_dtype, int64_t N_local, int c, int kk,
                                   const uint8_t *mask, const double *Y, int n, double *sums) {
    if (!ctx || !X || !mask || !Y || !sums) return CP_ERR_ARG;
    if (N_local <= 0 || c <= 0 || kk <= 0 || n <= 0) return cp_set_error(ctx, CP_ERR_ARG, "refit shard: bad sizes");
    if (x_dtype != CP_F32 && x_dtype != CP_F64) return cp_set_error(ctx, CP_ERR_ARG, "refit shard: bad dtype");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    ShardDims d;
    CP_TRY(shard_dims(ctx, c, kk, mask, n, d));
    const int RB = 64;
    const int rows_per_block = int((N_local + RB - 1) / RB);
    CP_TRY(cp_arena_reserve(ctx, size_t(RB) * size_t(d.p_pad + d.n_pad) * 8 + size_t(d.kept) * 4 + (1 << 16)));
    double *part_x = cp_arena_take_t<double>(ctx, size_t(RB) * d.p_pad);
    double *part_y = cp_arena_take_t<double>(ctx, size_t(RB) * d.n_pad);
    int *dchan = cp_arena_take_t<int>(ctx, d.kept);
    if (!part_x || !part_y || !dchan) return cp_set_error(ctx, CP_ERR_NOMEM, "refit shard: arena");
    CP_HIP(ctx, hipMemcpyAsync(dchan, d.chan.data(), size_t(d.kept) * 4, hipMemcpyHostToDevice, ctx->stream));
    CP_HIP(ctx, hipMemsetAsync(sums, 0, size_t(d.p_pad + d.n_pad) * 8, ctx->stream));
    const int gx = (d.p + RT - 1) / RT, gy = (n + RT - 1) / RT;
    dim3 gs(gx + gy, RB);
    if (x_dtype == CP_F32)
        k_colsum_xy<float><<<gs, RT, 0, ctx->stream>>>(static_cast<const float *>(X), Y, N_local, c, kk, n, dchan, d.p, gx,
                                                       rows_per_block, part_x, d.p_pad, part_y, d.n_pad);
    else
        k_colsum_xy<double><<<gs, RT, 0, ctx->stream>>>(static_cast<const double *>(X), Y, N_local, c, kk, n, dchan, d.p,
                                                        gx, rows_per_block, part_x, d.p_pad, part_y, d.n_pad);
    CP_LAUNCH_CHECK(ctx);
    k_mean_finish_xy<<<gx + gy, RT, 0, ctx->stream>>>(part_x, d.p_pad, d.p, part_y, d.n_pad, n, gx, RB, 1.0, sums,
                                                      sums + d.p_pad);
    CP_LAUNCH_CHECK(ctx);
    CP_HIP(ctx, cp_stream_wait(ctx));  // the caller's collective runs on another stream
    return CP_OK;
}

extern "C" int cp_refit_shard_gram(cp_ctx *ctx, const void *X, int x_dtype, int64_t N_local, int c, int kk,
                                   const uint8_t *mask, const double *Y, int n, int64_t N_total, const double *sums,
                                   double *gram) {
    if (!ctx || !X || !mask || !Y || !sums || !gram) return CP_ERR_ARG;
    if (N_local <= 0 || N_total < N_local || c <= 0 || kk <= 0 || n <= 0)
        return cp_set_error(ctx, CP_ERR_ARG, "refit shard: bad sizes");
    if (x_dtype != CP_F32 && x_dtype != CP_F64) return cp_set_error(ctx, CP_ERR_ARG, "refit shard: bad dtype");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    ShardDims d;
    CP_TRY(shard_dims(ctx, c, kk, mask, n, d));
    const int p_pad = d.p_pad, n_pad = d.n_pad;
    const int64_t N_pad = int64_t(cp_align_up(size_t(N_local), 16));
    const size_t ws = std::max(cp_gemm_tn_workspace(ctx, p_pad, p_pad, int(N_pad), CP_TRI_LOWER_MIRROR),
                               cp_gemm_tn_workspace(ctx, p_pad, n_pad, int(N_pad), CP_TRI_NONE));
    CP_TRY(cp_arena_reserve(ctx, size_t(N_pad) * size_t(p_pad + n_pad) * 8 + size_t(p_pad + n_pad) * 8 +
                                     size_t(d.kept) * 4 + ws + (1 << 16)));
    double *Xs = cp_arena_take_t<double>(ctx, size_t(N_pad) * p_pad);
    double *Yc = cp_arena_take_t<double>(ctx, size_t(N_pad) * n_pad);
    double *means = cp_arena_take_t<double>(ctx, size_t(p_pad + n_pad));
    int *dchan = cp_arena_take_t<int>(ctx, d.kept);
    if (!Xs || !Yc || !means || !dchan) return cp_set_error(ctx, CP_ERR_NOMEM, "refit shard: arena");
    CP_HIP(ctx, hipMemcpyAsync(dchan, d.chan.data(), size_t(d.kept) * 4, hipMemcpyHostToDevice, ctx->stream));
    k_scale_vec<<<(p_pad + n_pad + RT - 1) / RT, RT, 0, ctx->stream>>>(sums, 1.0 / double(N_total), p_pad + n_pad, means);
    CP_LAUNCH_CHECK(ctx);
    if (x_dtype == CP_F32)
        k_gather_center_xy<float><<<unsigned(N_pad), RT, 0, ctx->stream>>>(static_cast<const float *>(X), Y, N_local, c, kk,
                                                                           n, dchan, d.p, p_pad, n_pad, means,
                                                                           means + p_pad, Xs, Yc);
    else
        k_gather_center_xy<double><<<unsigned(N_pad), RT, 0, ctx->stream>>>(static_cast<const double *>(X), Y, N_local, c,
                                                                            kk, n, dchan, d.p, p_pad, n_pad, means,
                                                                            means + p_pad, Xs, Yc);
    CP_LAUNCH_CHECK(ctx);
    double *Gd = gram, *Rd = gram + size_t(p_pad) * p_pad;
    ctx->gemm_tag = CP_GEMM_REFIT_GRAM;
    ctx->gemm_mark = nullptr;
    CP_TRY(cp_gemm_tn_f64(ctx, p_pad, p_pad, int(N_pad), 1.0, Xs, p_pad, Xs, p_pad, 0.0, Gd, p_pad, CP_TRI_LOWER_MIRROR));
    ctx->gemm_tag = CP_GEMM_REFIT_XTY;
    CP_TRY(cp_gemm_tn_f64(ctx, p_pad, n_pad, int(N_pad), 1.0, Xs, p_pad, Yc, n_pad, 0.0, Rd, n_pad, CP_TRI_NONE));
    CP_HIP(ctx, cp_stream_wait(ctx));
    return CP_OK;
}

extern "C" int cp_refit_shard_solve(cp_ctx *ctx, int kept, int kk, int n, int64_t N_total, double ridge,
                                    const double *sums, const double *gram, double *W_out, double *b_out,
                                    cp_refit_info *info) {
    if (!ctx || !sums || !gram || !W_out || !b_out || !info) return CP_ERR_ARG;
    if (kept <= 0 || kk <= 0 || n <= 0 || N_total <= 0 || ridge < 0)
        return cp_set_error(ctx, CP_ERR_ARG, "refit shard: bad sizes");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    const int p = kept * kk;
    const int p_pad = int(cp_align_up(size_t(p), NB)), n_pad = int(cp_align_up(size_t(n), 128));
    const int nblk = p_pad / NB;
    const size_t g_b = size_t(p_pad) * p_pad * 8, r_b = size_t(p_pad) * n_pad * 8, ti_b = size_t(nblk) * NB * NB * 8;
    const size_t ws = cp_gemm_tn_workspace(ctx, p_pad, n_pad, p_pad, CP_TRI_NONE);
    CP_TRY(cp_arena_reserve(ctx, 4 * g_b + 4 * r_b + 2 * ti_b + size_t(p_pad) * 8 * 3 + size_t(n_pad) * 8 + ws +
                                     size_t(chol_info_count(nblk)) * 4 + (1 << 16)));
    double *G = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *G0 = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *Lt = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *Uf = cp_arena_take_t<double>(ctx, size_t(p_pad) * p_pad);
    double *Yt = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
    double *Rm = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
    double *R2 = cp_arena_take_t<double>(ctx, size_t(p_pad) * n_pad);
    double *TI = cp_arena_take_t<double>(ctx, size_t(nblk) * NB * NB);
    double *TIT = cp_arena_take_t<double>(ctx, size_t(nblk) * NB * NB);
    double *means = cp_arena_take_t<double>(ctx, size_t(p_pad + n_pad));
    double *dg0 = cp_arena_take_t<double>(ctx, p_pad);
    double *gmax = cp_arena_take_t<double>(ctx, 8);
    int *dinfo = cp_arena_take_t<int>(ctx, chol_info_count(nblk) + 16);
    if (!G || !G0 || !Lt || !Uf || !Yt || !Rm || !R2 || !TI || !TIT || !means || !dg0 || !gmax || !dinfo)
        return cp_set_error(ctx, CP_ERR_NOMEM, "refit shard: arena");
    CP_TRY(cp_pinned_reserve(ctx, 64));
    int *info_host = reinterpret_cast<int *>(ctx->pinned);
    cp_stage_begin(ctx);
    k_scale_vec<<<(p_pad + n_pad + RT - 1) / RT, RT, 0, ctx->stream>>>(sums, 1.0 / double(N_total), p_pad + n_pad, means);
    CP_LAUNCH_CHECK(ctx);
    // the "normal equations" of this path: the all-reduced Gram and right-hand side, diagonal prepared as usual
    auto normal_equations = [&](double *Gd, double *Rd, bool) -> int {
        CP_HIP(ctx, hipMemcpyAsync(Gd, gram, g_b, hipMemcpyDeviceToDevice, ctx->stream));
        CP_HIP(ctx, hipMemcpyAsync(Rd, gram + size_t(p_pad) * p_pad, r_b, hipMemcpyDeviceToDevice, ctx->stream));
        k_diag_prepare<<<1, 256, 0, ctx->stream>>>(Gd, p_pad, p, p_pad, ridge, dg0, gmax, dinfo, chol_info_count(nblk));
        CP_LAUNCH_CHECK(ctx);
        return CP_OK;
    };
    RefitSolve rs{G,  G0, Lt, Uf, Yt, Rm, R2, TI, TIT, means, means + p_pad, dg0, gmax, dinfo, p, p_pad, n, n_pad, nblk,
                  N_total, ridge, W_out, b_out, info_host, nullptr, nullptr};
    return refit_solve_tail(ctx, rs, normal_equations, info);
}
