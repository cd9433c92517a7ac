// Projector onto the r leading eigenvectors of a symmetric matrix WITHOUT its eigen-decomposition:
//     P_r = (I + sign(A - sigma I)) / 2      for any sigma strictly between lambda_{r+1} and lambda_r,
// sign() by the Newton-Schulz iteration X <- X (3 I - X^2) / 2 -- two n x n products per step, nothing else.
// Used by cp_itq_iterate (refit.hip) for the rank-truncation step of ITQ_decompose (lib/decompose.py:218-220: everything
// the alternation needs from the SVD of X is X V_r V_r^T): from the third alternation on the spectrum moves a few per
// cent per alternation and has a wide gap behind lambda_r (lambda_{r+1} / lambda_r ~ 0.5 at lambda = 0.1, ~ 0.13 at
// lambda = 1 in the conv3-size golden), so the threshold of the previous alternation, scaled by the ratio of the traces,
// still separates the same r eigenvalues.  That is CHECKED, not assumed: the iteration has converged when
// |I - X^2|_F < 1e-7 before the last step (quadratic: the step after it is at rounding level), and trace(P) must then be
// the integer r -- exactly r eigenvalues above sigma, so P IS the projector the reference's SVD truncation defines.  A
// wrong count moves sigma (bracketing, geometric steps of 1.3) and repeats; the caller falls back to the Jacobi sweeps
// (svd_jacobi.hip) if no threshold is found.  14-17 steps of two launches against 7 warm-started sweeps of 31 (n = 256)
// or 63 (n = 512) dependent launches.
//
// k_ns_gemm: C = alpha A^T B + beta D on square n_pad x n_pad operands (n_pad a multiple of 32).  One 16 x 16 MFMA tile per
// wave, four waves = a 32 x 32 tile per workgroup, so that n_pad = 256 fills 64 CUs and 512 all of them: with the
// 128 x 128 tiles of gemm_f64.hip these products are four to sixteen workgroups of 11-22 us each.
#include "cp_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef double v4f64s __attribute__((ext_vector_type(4)));
constexpr int NT = 256;

// off[i] = sum_{j != i} |(a_ij + a_ji) / 2|, dg[i] = a_ii
__global__ void __launch_bounds__(NT) k_ns_rowstat(const double *__restrict__ A, int np_, double *__restrict__ off,
                                                   double *__restrict__ dg) {
    __shared__ double red[4];
    const int i = blockIdx.x;
    double s = 0;
    for (int j = threadIdx.x; j < np_; j += NT)
        if (j != i) s += fabs(0.5 * (A[size_t(i) * np_ + j] + A[size_t(j) * np_ + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        off[i] = red[0] + red[1] + red[2] + red[3];
        dg[i] = A[size_t(i) * np_ + i];
    }
}

// X = (sym(A) - sigma I) / s with sigma = sigma_abs >= 0 ? sigma_abs : sigma_rel * trace(A) and s = the infinity norm of
// the shifted matrix (>= its spectral radius: every eigenvalue of X lies in [-1, 1]).  info = {trace, sigma, s}.
__global__ void __launch_bounds__(NT) k_ns_start(const double *__restrict__ A, int np_, const double *__restrict__ off,
                                                 const double *__restrict__ dg, double sigma_abs, double sigma_rel,
                                                 double *__restrict__ X, double *__restrict__ info) {
    __shared__ double red[4], sh_sigma, sh_s;
    double t = 0;
    for (int j = threadIdx.x; j < np_; j += NT) t += dg[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) sh_sigma = sigma_abs >= 0.0 ? sigma_abs : sigma_rel * (red[0] + red[1] + red[2] + red[3]);
    const double tr = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const double sigma = sh_sigma;
    double m = 0;
    for (int j = threadIdx.x; j < np_; j += NT) m = fmax(m, fabs(dg[j] - sigma) + off[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) sh_s = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    const double s = sh_s, inv = s > 0.0 ? 1.0 / s : 0.0;
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < np_; j += NT) {
        const double a = 0.5 * (A[size_t(i) * np_ + j] + A[size_t(j) * np_ + i]);
        X[size_t(i) * np_ + j] = (a - (i == j ? sigma : 0.0)) * inv;
    }
    if (i == 0 && threadIdx.x == 0) {
        info[0] = tr;
        info[1] = sigma;
        info[2] = s;
    }
}

// C = alpha A^T B + beta D (D may be null with beta = 0); RES: also part[workgroup] = sum over the tile of
// (delta_ij - A^T B)^2 -- the squared Frobenius distance of X^2 from the identity, summed on the host in workgroup order.
// (register budget of two waves per SIMD: the compiler then keeps the accumulator in VGPRs -- the AccVGPR form of
//  v_mfma_f64_16x16x4_f64 it picks for a one-wave-per-SIMD kernel issues 1.65 x slower, profiles/r04_gemm_probe.md)
template <bool RES>
__global__ void __launch_bounds__(NT, 2) k_ns_gemm(const double *__restrict__ A, const double *__restrict__ B, int np_,
                                                double alpha, double beta, const double *__restrict__ D,
                                                double *__restrict__ C, double *__restrict__ part) {
    const int tiles = np_ / 32, tm = blockIdx.x / tiles, tn = blockIdx.x % tiles;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fk = lane >> 4, fi = lane & 15;
    const int m0 = tm * 32 + (wave >> 1) * 16, n0 = tn * 32 + (wave & 1) * 16;
    const double *ap = A + size_t(fk) * np_ + m0 + fi, *bp = B + size_t(fk) * np_ + n0 + fi;
    v4f64s acc = {0., 0., 0., 0.};
    constexpr int U = 8;                      // k-steps (of 4 rows) whose loads are in flight together
    double a[2][U], b[2][U];
    auto fetch = [&](int k0, double *ad, double *bd) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            ad[u] = ap[size_t(k0 + 4 * u) * np_];
            bd[u] = bp[size_t(k0 + 4 * u) * np_];
        }
    };
    fetch(0, a[0], b[0]);                     // np_ is a multiple of 32 = 4 U
    for (int k0 = 0; k0 < np_; k0 += 8 * U) {
        if (k0 + 4 * U < np_) fetch(k0 + 4 * U, a[1], b[1]);
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0][u], b[0][u], acc, 0, 0, 0);
        if (k0 + 8 * U < np_) fetch(k0 + 8 * U, a[0], b[0]);
        if (k0 + 4 * U < np_) {
#pragma unroll
            for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1][u], b[1][u], acc, 0, 0, 0);
        }
    }
    double r2 = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + fk + 4 * r, col = n0 + fi;
        const double p = acc[r];
        if (RES) {
            const double d = (row == col ? 1.0 : 0.0) - p;
            r2 = fma(d, d, r2);
        }
        double v = alpha * p;
        if (beta != 0.0) v = fma(beta, D[size_t(row) * np_ + col], v);
        C[size_t(row) * np_ + col] = v;
    }
    if (RES) {
        __shared__ double red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) r2 += __shfl_xor(r2, o, 64);
        if (lane == 0) red[wave] = r2;
        __syncthreads();
        if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

// P = (I + sym(X)) / 2, dvec[i] = P_ii
__global__ void __launch_bounds__(NT) k_ns_finish(const double *__restrict__ X, int np_, double *__restrict__ P,
                                                  double *__restrict__ dvec) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < np_; j += NT) {
        const double v = 0.5 * (0.5 * (X[size_t(i) * np_ + j] + X[size_t(j) * np_ + i]) + (i == j ? 1.0 : 0.0));
        P[size_t(i) * np_ + j] = v;
        if (i == j) dvec[i] = v;
    }
}

constexpr int NS_MAX_STEPS = 64;
constexpr double NS_RES_DONE = 1e-7;   // |I - X^2|_F before the last step: the step after it leaves ~ its square

}  // namespace

size_t cp_sign_workspace_doubles(int np_) {
    const size_t sq = size_t(np_) * np_, wgs = size_t(np_ / 32) * (np_ / 32);
    return 3 * sq + 3 * size_t(np_) + NS_MAX_STEPS * wgs + 64;
}

// A DEVICE [np_, np_] symmetric (zero beyond its live part), r = the number of eigenvalues wanted above the threshold.
// P DEVICE [np_, np_] <- the projector.  *found = false (and CP_OK): no threshold with exactly r eigenvalues above it was
// found / the iteration did not converge -- the caller decomposes A itself.  work: cp_sign_workspace_doubles(np_) doubles.
int cp_sign_projector(cp_ctx *ctx, const double *A, int np_, int r, SignTracker &tk, double *work, double *P, bool *found) {
    *found = false;
    if (np_ % 32 || r <= 0 || r >= np_ || !(tk.sigma_rel > 0.0)) return cp_set_error(ctx, CP_ERR_ARG, "sign_projector: bad arguments");
    const size_t sq = size_t(np_) * np_;
    const int wgs = (np_ / 32) * (np_ / 32);
    double *X0 = work, *X1 = X0 + sq, *Y = X1 + sq, *off = Y + sq, *dg = off + np_, *dvec = dg + np_;
    double *part = dvec + np_, *info = part + size_t(NS_MAX_STEPS) * wgs;
    const size_t host_doubles = size_t(NS_MAX_STEPS) * wgs + np_ + 8;
    CP_TRY(cp_pinned_reserve(ctx, host_doubles * 8 + 64));
    double *h = reinterpret_cast<double *>(ctx->pinned);
    k_ns_rowstat<<<np_, NT, 0, ctx->stream>>>(A, np_, off, dg);
    CP_LAUNCH_CHECK(ctx);
    double sigma_abs = -1.0, lo = 0.0, hi = 0.0;      // bracket of thresholds: lo has too many eigenvalues above it, hi too few
    for (int trial = 0; trial < 10; ++trial) {
        ++tk.trials;
        k_ns_start<<<np_, NT, 0, ctx->stream>>>(A, np_, off, dg, sigma_abs, tk.sigma_rel, X0, info);
        CP_LAUNCH_CHECK(ctx);
        double *Xc = X0, *Xn = X1;
        int done = 0, plan = std::min(NS_MAX_STEPS, std::max(4, tk.steps_plan)), conv_at = -1;
        double trace = 0, hinfo[3] = {0, 0, 0};
        while (true) {
            for (int it = done; it < plan; ++it) {
                k_ns_gemm<true><<<wgs, NT, 0, ctx->stream>>>(Xc, Xc, np_, 1.0, 0.0, nullptr, Y, part + size_t(it) * wgs);
                CP_LAUNCH_CHECK(ctx);
                k_ns_gemm<false><<<wgs, NT, 0, ctx->stream>>>(Xc, Y, np_, -0.5, 1.5, Xc, Xn, nullptr);
                CP_LAUNCH_CHECK(ctx);
                std::swap(Xc, Xn);
            }
            k_ns_finish<<<np_, NT, 0, ctx->stream>>>(Xc, np_, P, dvec);
            CP_LAUNCH_CHECK(ctx);
            const size_t nres = size_t(plan - done) * wgs;
            CP_HIP(ctx, hipMemcpyAsync(h, part + size_t(done) * wgs, nres * 8, hipMemcpyDeviceToHost, ctx->stream));
            CP_HIP(ctx, hipMemcpyAsync(h + nres, dvec, size_t(np_) * 8, hipMemcpyDeviceToHost, ctx->stream));
            CP_HIP(ctx, hipMemcpyAsync(h + nres + np_, info, 3 * 8, hipMemcpyDeviceToHost, ctx->stream));
            CP_HIP(ctx, cp_stream_wait(ctx));
            for (int it = done; it < plan && conv_at < 0; ++it) {
                double s = 0;
                for (int w = 0; w < wgs; ++w) s += h[size_t(it - done) * wgs + w];
                if (!(s == s)) return cp_set_error(ctx, CP_ERR_NUMERIC, "sign_projector: NaN in the iteration");
                if (std::sqrt(s) < NS_RES_DONE) conv_at = it;   // X_it was that close: X_{it+1} is converged
            }
            tk.steps += plan - done;
            done = plan;
            trace = 0;
            for (int i = 0; i < np_; ++i) trace += h[nres + i];
            memcpy(hinfo, h + nres + np_, sizeof(hinfo));
            if (conv_at >= 0 || plan >= NS_MAX_STEPS) break;
            plan = std::min(NS_MAX_STEPS, plan + 4);
        }
        const double sigma = hinfo[1];
        if (conv_at < 0) return CP_OK;                           // an eigenvalue sits (almost) on the threshold
        const long count = std::lround(trace);
        if (std::fabs(trace - double(count)) > 1e-6) return CP_OK;
        if (count == r) {
            tk.sigma_rel = sigma / hinfo[0];
            tk.steps_plan = conv_at + 2;                         // the steps it took + one in reserve
            *found = true;
            return CP_OK;
        }
        if (count > r) lo = sigma; else hi = sigma;
        if (lo > 0.0 && hi > 0.0) {
            if (hi <= lo * (1.0 + 1e-12)) return CP_OK;
            sigma_abs = std::sqrt(lo * hi);
        } else {
            sigma_abs = count > r ? sigma * 1.3 : sigma / 1.3;
        }
    }
    return CP_OK;
}

// Test hook (tests/test_gpu_parity.py): A HOST [n, n] symmetric, sigma_rel = threshold / trace to start from.
// P HOST [n, n]; out[0] = found, out[1] = Newton-Schulz steps, out[2] = thresholds tried.
extern "C" int cp_debug_sign_projector(cp_ctx *ctx, const double *A, int n, int r, double sigma_rel, double *P, int *out) {
    if (!ctx || !A || !P || !out || n <= 0) return CP_ERR_ARG;
    CP_HIP(ctx, hipSetDevice(ctx->device));
    const int np_ = int(cp_align_up(size_t(n), 128));
    const size_t sq = size_t(np_) * np_;
    CP_TRY(cp_arena_reserve(ctx, (2 * sq + cp_sign_workspace_doubles(np_)) * 8 + (1 << 16)));
    double *Ad = cp_arena_take_t<double>(ctx, sq), *Pd = cp_arena_take_t<double>(ctx, sq);
    double *work = cp_arena_take_t<double>(ctx, cp_sign_workspace_doubles(np_));
    if (!Ad || !Pd || !work) return cp_set_error(ctx, CP_ERR_NOMEM, "sign_projector: arena");
    CP_HIP(ctx, hipMemsetAsync(Ad, 0, sq * 8, ctx->stream));
    CP_HIP(ctx, hipMemcpy2DAsync(Ad, size_t(np_) * 8, A, size_t(n) * 8, size_t(n) * 8, size_t(n), hipMemcpyHostToDevice,
                                 ctx->stream));
    SignTracker tk;
    tk.sigma_rel = sigma_rel;
    bool found = false;
    CP_TRY(cp_sign_projector(ctx, Ad, np_, r, tk, work, Pd, &found));
    CP_HIP(ctx, hipMemcpy2DAsync(P, size_t(n) * 8, Pd, size_t(np_) * 8, size_t(n) * 8, size_t(n), hipMemcpyDeviceToHost,
                                 ctx->stream));
    CP_HIP(ctx, cp_stream_wait(ctx));
    out[0] = found ? 1 : 0;
    out[1] = int(tk.steps);
    out[2] = tk.trials;
    return CP_OK;
}
