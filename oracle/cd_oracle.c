/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the arithmetic on the
 * channel-pruning hot path.  Never linked into, imported by, or called from the product
 * (channel-pruning_amd/); only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker.
 *
 * The arithmetic of the reference's path lives in third-party code that is NOT under
 * /root/reference (versions unpinned by the reference, README.md:46); this file restates
 * the published algorithms of the versions installed in this image:
 *   scikit-learn 1.7.2  sklearn/linear_model/_cd_fast.pyx:101-273   enet_coordinate_descent
 *                       sklearn/linear_model/_cd_fast.pyx:564-737   enet_coordinate_descent_gram
 *                       sklearn/utils/_random.pxd:20-35             our_rand_r (xorshift32)
 *                       sklearn/linear_model/_base.py:108-205       _preprocess_data (centring)
 * and the reference's own operand construction:
 *   lib/decompose.py:425-437   Z = matmul(reX, reW2).reshape(c,-1).T ; reY = Y[samples].reshape(-1)
 *   lib/net.py:629-657         extract_XY: zero-padded k x k window copy at sampled points
 *   lib/net.py:1702,1707,1720  rollaxis -> [N,c,k,k]; Y = feats - bias; relu(newX)
 *
 * Pinned (tests/test_oracle.py, oracle/validate_oracle.py): data-form CD against
 * sklearn's own Lasso (same seeds -> same n_iter and zero pattern), Gram-form CD against
 * sklearn's enet_coordinate_descent_gram, Z/Gram against numpy, and the whole pipeline
 * against golden vectors produced by the unmodified reference (tests/golden/).
 *
 * Build: gcc -O2 -march=x86-64-v3 -ffp-contract=off -shared -fPIC cd_oracle.c -o _build/libcporacle.so -lm
 * (-ffp-contract=off: every fused multiply-add below is an explicit fma(), so the HIP
 *  kernels can reproduce the same rounding sequence.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define CPO_RAND_R_MAX 2147483647u

/* sklearn/utils/_random.pxd:20-35 */
uint32_t cpo_rand_r(uint32_t *seed) {
    if (*seed == 0) *seed = 1;
    *seed ^= (uint32_t)(*seed << 13);
    *seed ^= (uint32_t)(*seed >> 17);
    *seed ^= (uint32_t)(*seed << 5);
    return *seed % (CPO_RAND_R_MAX + 1u);
}

/* _cd_fast.pyx:30-32 */
static inline uint32_t rand_int(uint32_t end, uint32_t *state) { return cpo_rand_r(state) % end; }

static inline double fsign(double f) { return f == 0 ? 0.0 : (f > 0 ? 1.0 : -1.0); }
static inline double fmax_(double x, double y) { return x > y ? x : y; }

/* first `count` coordinates visited by a fit seeded with `seed` (spec for the device RNG) */
void cpo_coord_sequence(uint32_t seed, uint32_t n_features, int64_t count, int32_t *out) {
    uint32_t s = seed;
    for (int64_t i = 0; i < count; ++i) out[i] = (int32_t)rand_int(n_features, &s);
}

/* 4-accumulator dot: same flop count/streaming pattern as the BLAS ddot sklearn calls. */
static double dot4(int64_t n, const double *a, const double *b) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        s0 += a[i] * b[i];
        s1 += a[i + 1] * b[i + 1];
        s2 += a[i + 2] * b[i + 2];
        s3 += a[i + 3] * b[i + 3];
    }
    for (; i < n; ++i) s0 += a[i] * b[i];
    return (s0 + s1) + (s2 + s3);
}
static void axpy(int64_t n, double alpha, const double *x, double *y) {
    for (int64_t i = 0; i < n; ++i) y[i] += alpha * x[i];
}

/*
 * Data form, _cd_fast.pyx:101-273.  X is Fortran-ordered [n_samples, n_features] and
 * already centred, y centred; w is the warm start and the result.  Returns n_iter
 * (= the "n_iter + 1" sklearn returns).
 */
int cpo_enet_cd_data(double *w, double alpha, double beta, const double *X, const double *y,
                     int64_t n_samples, int32_t n_features, int32_t max_iter, double tol,
                     uint32_t seed, int32_t random, double *gap_out, double *tol_out) {
    double *norm_cols = (double *)malloc(sizeof(double) * n_features);
    double *R = (double *)malloc(sizeof(double) * n_samples);
    double *XtA = (double *)malloc(sizeof(double) * n_features);
    double gap = tol + 1.0, d_w_tol = tol;
    uint32_t state = seed;
    int32_t n_iter = 0;
    for (int32_t j = 0; j < n_features; ++j)  /* np.square(X).sum(axis=0) */
        norm_cols[j] = dot4(n_samples, X + (int64_t)j * n_samples, X + (int64_t)j * n_samples);
    memcpy(R, y, sizeof(double) * n_samples);              /* R = y - X w */
    for (int32_t j = 0; j < n_features; ++j)
        if (w[j] != 0.0) axpy(n_samples, -w[j], X + (int64_t)j * n_samples, R);
    tol *= dot4(n_samples, y, y);
    for (n_iter = 0; n_iter < max_iter; ++n_iter) {
        double w_max = 0.0, d_w_max = 0.0;
        for (int32_t f = 0; f < n_features; ++f) {
            uint32_t ii = random ? rand_int((uint32_t)n_features, &state) : (uint32_t)f;
            if (norm_cols[ii] == 0.0) continue;
            const double *col = X + (int64_t)ii * n_samples;
            double w_ii = w[ii];
            if (w_ii != 0.0) axpy(n_samples, w_ii, col, R);
            double tmp = dot4(n_samples, col, R);
            w[ii] = fsign(tmp) * fmax_(fabs(tmp) - alpha, 0) / (norm_cols[ii] + beta);
            if (w[ii] != 0.0) axpy(n_samples, -w[ii], col, R);
            double d_w_ii = fabs(w[ii] - w_ii);
            d_w_max = fmax_(d_w_max, d_w_ii);
            w_max = fmax_(w_max, fabs(w[ii]));
        }
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            double dual_norm = 0.0, R_norm2, w_norm2, l1 = 0.0, const_, A_norm2;
            for (int32_t j = 0; j < n_features; ++j) {
                XtA[j] = dot4(n_samples, X + (int64_t)j * n_samples, R) - beta * w[j];
                if (fabs(XtA[j]) > dual_norm) dual_norm = fabs(XtA[j]);
                l1 += fabs(w[j]);
            }
            R_norm2 = dot4(n_samples, R, R);
            w_norm2 = dot4(n_features, w, w);
            if (dual_norm > alpha) {
                const_ = alpha / dual_norm;
                A_norm2 = R_norm2 * (const_ * const_);
                gap = 0.5 * (R_norm2 + A_norm2);
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += alpha * l1 - const_ * dot4(n_samples, R, y) +
                   0.5 * beta * (1 + const_ * const_) * w_norm2;
            if (gap < tol) break;
        }
    }
    if (n_iter == max_iter) n_iter = max_iter - 1; /* for/else: no break */
    *gap_out = gap;
    *tol_out = tol;
    free(norm_cols);
    free(R);
    free(XtA);
    return n_iter + 1;
}

/*
 * Gram form, _cd_fast.pyx:564-737.  Q = Xc^T Xc (C-ordered, symmetric), q = Xc^T yc,
 * y_norm2 = yc^T yc.  The two _axpy calls of the original are written as explicit fma()
 * per element (what an FMA-capable BLAS daxpy does) so that the device kernel
 * (channel-pruning_amd/csrc/cd_gram.hip) can reproduce w bit-for-bit.
 * recip != 0 replaces "/ (Q[ii,ii] + beta)" by "* (1 / (Q[ii,ii] + beta))" (a <=1 ulp
 * deviation offered by the device kernel as a latency option; default 0 = faithful).
 * bit 1 of recip (value 2) selects the "delta" form: the two daxpy H -= w_ii Q[ii]; H += w_new Q[ii]
 * become one, H += (w_new - w_ii) Q[ii], while tmp still uses H[ii] - w_ii Q[ii,ii] (again a
 * rounding-level deviation offered as a device option).
 * stats_out (may be NULL) = {gap, tol_scaled, q_dot_w, dual_norm_XtA, R_norm2}.
 */
int cpo_enet_cd_gram(double *w, double alpha, double beta, const double *Q, const double *q,
                     double y_norm2, int32_t n_features, int32_t max_iter, double tol,
                     uint32_t seed, int32_t random, int32_t recip, double *stats_out) {
    double *H = (double *)calloc(n_features, sizeof(double));
    double gap = tol + 1.0, d_w_tol = tol;
    double q_dot_w = 0, dual_norm = 0, R_norm2 = 0;
    uint32_t state = seed;
    int32_t n_iter = 0;
    /* H = np.dot(Q, w): accumulate column by column in index order with fma */
    for (int32_t j = 0; j < n_features; ++j)
        if (w[j] != 0.0)
            for (int32_t i = 0; i < n_features; ++i)
                H[i] = fma(w[j], Q[(int64_t)j * n_features + i], H[i]);
    tol = tol * y_norm2;
    for (n_iter = 0; n_iter < max_iter; ++n_iter) {
        double w_max = 0.0, d_w_max = 0.0;
        for (int32_t f = 0; f < n_features; ++f) {
            uint32_t ii = random ? rand_int((uint32_t)n_features, &state) : (uint32_t)f;
            const double *Qi = Q + (int64_t)ii * n_features;
            if (Qi[ii] == 0.0) continue;
            double w_ii = w[ii];
            const int delta = (recip & 2) != 0;
            double tmp;
            if (delta) {
                tmp = q[ii] - fma(-w_ii, Qi[ii], H[ii]);
            } else {
                if (w_ii != 0.0)
                    for (int32_t i = 0; i < n_features; ++i) H[i] = fma(-w_ii, Qi[i], H[i]);
                tmp = q[ii] - H[ii];
            }
            double thr = fsign(tmp) * fmax_(fabs(tmp) - alpha, 0);
            w[ii] = (recip & 1) ? thr * (1.0 / (Qi[ii] + beta)) : thr / (Qi[ii] + beta);
            if (delta) {
                const double d = w[ii] - w_ii;
                if (d != 0.0)
                    for (int32_t i = 0; i < n_features; ++i) H[i] = fma(d, Qi[i], H[i]);
            } else if (w[ii] != 0.0)
                for (int32_t i = 0; i < n_features; ++i) H[i] = fma(w[ii], Qi[i], H[i]);
            double d_w_ii = fabs(w[ii] - w_ii);
            if (d_w_ii > d_w_max) d_w_max = d_w_ii;
            if (fabs(w[ii]) > w_max) w_max = fabs(w[ii]);
        }
        if (w_max == 0.0 || d_w_max / w_max < d_w_tol || n_iter == max_iter - 1) {
            double tmp = 0.0, w_norm2 = 0.0, l1 = 0.0, const_, A_norm2;
            q_dot_w = 0.0;
            dual_norm = 0.0;
            for (int32_t i = 0; i < n_features; ++i) {
                double xta = q[i] - H[i] - beta * w[i];
                q_dot_w += w[i] * q[i];
                if (fabs(xta) > dual_norm) dual_norm = fabs(xta);
                tmp += w[i] * H[i];
                w_norm2 += w[i] * w[i];
                l1 += fabs(w[i]);
            }
            R_norm2 = y_norm2 + tmp - 2.0 * q_dot_w;
            if (dual_norm > alpha) {
                const_ = alpha / dual_norm;
                A_norm2 = R_norm2 * (const_ * const_);
                gap = 0.5 * (R_norm2 + A_norm2);
            } else {
                const_ = 1.0;
                gap = R_norm2;
            }
            gap += alpha * l1 - const_ * y_norm2 + const_ * q_dot_w +
                   0.5 * beta * (1 + const_ * const_) * w_norm2;
            if (gap < tol) break;
        }
    }
    if (n_iter == max_iter) n_iter = max_iter - 1;
    if (stats_out) {
        stats_out[0] = gap;
        stats_out[1] = tol;
        stats_out[2] = q_dot_w;
        stats_out[3] = dual_norm;
        stats_out[4] = R_norm2;
    }
    free(H);
    return n_iter + 1;
}

/*
 * LASSO operands, lib/decompose.py:425-437 + sklearn centring (_base.py:108-205):
 *   Z[(s,j), i] = sum_t X[samples[s], i, t] * W2[j, i, t]          (M = S*n rows, c columns)
 *   y[(s,j)]    = Y[samples[s], j]
 * Outputs (all float64): Zc Fortran-ordered [M, c] centred (may be NULL),
 * yc [M] centred (may be NULL), Q = Zc^T Zc [c,c], q = Zc^T yc [c], stats = {yc^T yc, mean(y)},
 * zmean [c].  X is [N, c, kk] float64, W2 [n, c, kk] float64.
 */
void cpo_lasso_operands(const double *X, const double *W2, const double *Y, const int64_t *samples,
                        int64_t S, int32_t c, int32_t n, int32_t kk, double *Zc, double *yc,
                        double *Q, double *q, double *stats, double *zmean) {
    int64_t M = S * (int64_t)n;
    double *Z = Zc ? Zc : (double *)malloc(sizeof(double) * M * c);
    double *y = yc ? yc : (double *)malloc(sizeof(double) * M);
    for (int32_t i = 0; i < c; ++i) {
        double *col = Z + (int64_t)i * M;
        double sum = 0.0;
        for (int64_t s = 0; s < S; ++s) {
            const double *x = X + (samples[s] * c + i) * kk;
            for (int32_t j = 0; j < n; ++j) {
                const double *wv = W2 + ((int64_t)j * c + i) * kk;
                double acc = 0.0;
                for (int32_t t = 0; t < kk; ++t) acc += x[t] * wv[t];
                col[s * n + j] = acc;
                sum += acc;
            }
        }
        double mu = sum / (double)M;
        zmean[i] = mu;
        for (int64_t r = 0; r < M; ++r) col[r] -= mu;
    }
    double ysum = 0.0;
    for (int64_t s = 0; s < S; ++s)
        for (int32_t j = 0; j < n; ++j) {
            y[s * n + j] = Y[samples[s] * n + j];
            ysum += y[s * n + j];
        }
    double ymean = ysum / (double)M;
    for (int64_t r = 0; r < M; ++r) y[r] -= ymean;
    stats[0] = dot4(M, y, y);
    stats[1] = ymean;
    for (int32_t i = 0; i < c; ++i) {
        q[i] = dot4(M, Z + (int64_t)i * M, y);
        for (int32_t i2 = 0; i2 <= i; ++i2) {
            double v = dot4(M, Z + (int64_t)i * M, Z + (int64_t)i2 * M);
            Q[(int64_t)i * c + i2] = v;
            Q[(int64_t)i2 * c + i] = v;
        }
    }
    if (!Zc) free(Z);
    if (!yc) free(y);
}

/*
 * Sampled-point im2col, lib/net.py:629-657 (non-gw1 branch) followed by net.py:1702 and
 * the VGG ReLU of net.py:1720.  One call = one batch: fmap [B, C, H, W] float32 (the bottom
 * blob before padding), points (xs[p], ys[p]) in TOP coordinates.  Row order produced:
 * [point][image]; each row is a [C, k, k] patch (the reference's feats[N*k*k, C] viewed as
 * [N, k, k, C] then rollaxis(3,1)).  Window rows x*stride .. x*stride+k-1 of the padded map.
 */
void cpo_patch_gather(const float *fmap, int32_t B, int32_t C, int32_t H, int32_t W, const int32_t *xs,
                      const int32_t *ys, int32_t P, int32_t k, int32_t pad, int32_t stride, int32_t relu,
                      float *out) {
    for (int32_t p = 0; p < P; ++p)
        for (int32_t b = 0; b < B; ++b)
            for (int32_t ch = 0; ch < C; ++ch)
                for (int32_t dh = 0; dh < k; ++dh)
                    for (int32_t dw = 0; dw < k; ++dw) {
                        int32_t hh = xs[p] * stride + dh - pad, ww = ys[p] * stride + dw - pad;
                        float v = 0.0f;
                        if (hh >= 0 && hh < H && ww >= 0 && ww < W)
                            v = fmap[(((int64_t)b * C + ch) * H + hh) * W + ww];
                        if (relu && v < 0.0f) v = 0.0f;
                        out[((((int64_t)p * B + b) * C + ch) * k + dh) * k + dw] = v;
                    }
}

/* Y = feats - bias (+ resY), lib/net.py:1707,1722: float32 blobs widened to float64 first. */
void cpo_assemble_y(const float *feats, const float *bias, const double *resY, int64_t N, int32_t n,
                    double *Y) {
    for (int64_t r = 0; r < N; ++r)
        for (int32_t j = 0; j < n; ++j) {
            double v = (double)feats[r * n + j] - (double)bias[j];
            if (resY) v += resY[r * n + j];
            Y[r * n + j] = v;
        }
}
