/*
 * cpmi355.h -- C ABI of libcpmi355.so: the MI355X (gfx950) implementation of the
 * channel-pruning hot path of ethanhe42/channel-pruning.
 *
 * The reference has NO FFI layer for this path: it is plain Python calling
 * scikit-learn / SciPy (lib/decompose.py:386-669, lib/net.py:534-684,1685-1735).  This
 * header is therefore the boundary a maintainer would bind with ctypes from the bodies of
 * those functions (binding shown in INTEGRATION.md; shipped binding:
 * channel-pruning_amd/cpmi355/capi.py).  Each entry point cites what it replaces.
 *
 * Conventions
 *  - extern "C", every function returns int: CP_OK (0) or a negative CP_ERR_* code; no
 *    exception crosses the boundary.  cp_last_error(ctx) gives the detail string.
 *  - cp_ctx binds one device and one HIP stream; one ctx per thread/process.  Create it
 *    AFTER fork() (the reference likewise creates its GPU context inside the child,
 *    lib/net.py:55-58, lib/worker.py:33).
 *  - Pointers marked DEVICE are device addresses valid on the ctx's device (hipMalloc,
 *    cp_malloc, or a torch tensor's data_ptr()); pointers marked HOST are ordinary host
 *    memory.  Inputs are never modified.  Calls are asynchronous on the ctx stream
 *    unless they return HOST outputs, in which case they synchronise the stream.
 *  - Row-major (C order) everywhere; dtype codes CP_F32 / CP_F64.
 */
#ifndef CPMI355_H
#define CPMI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CP_VERSION 100 /* 0.1.0 */

#define CP_OK 0
#define CP_ERR_ARG (-1)         /* bad argument (null pointer, size, dtype) */
#define CP_ERR_HIP (-2)         /* a HIP runtime call failed */
#define CP_ERR_NOMEM (-3)       /* device allocation failed */
#define CP_ERR_UNSUPPORTED (-4) /* shape outside what the kernels support */
#define CP_ERR_NUMERIC (-5)     /* factorisation broke down and no fallback applied */
#define CP_ERR_NODEVICE (-6)    /* no gfx950 device visible */

#define CP_F32 0
#define CP_F64 1

typedef struct cp_ctx cp_ctx;

/* ---- library / context ---------------------------------------------------------- */
int cp_version(void);
const char *cp_strerror(int code);
const char *cp_last_error(const cp_ctx *ctx);
int cp_device_count(int *count);
int cp_ctx_create(int device, cp_ctx **out);
/* The same with an explicit HIP stream priority for the context's stream (0 = default, < 0 = higher, > 0 = lower; clamped to
 * the device's range).  cp_ctx_create() uses the environment's CP_CTX_PRIORITY (default 0). */
int cp_ctx_create_priority(int device, int priority, cp_ctx **out);
int cp_ctx_destroy(cp_ctx *ctx);
/* a context with its own workspace that runs on `of`'s stream (no stream / hardware queue of its own): the per-job
 * contexts of cp_prune_layers.  Destroy it before `of`. */
int cp_ctx_create_sibling(cp_ctx *of, cp_ctx **out);
/* run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL restores the ctx's own stream. */
int cp_ctx_set_stream(cp_ctx *ctx, void *hip_stream);
int cp_sync(cp_ctx *ctx);

/* ---- device memory helpers (so a ctypes-only host needs nothing else) ------------ */
int cp_malloc(cp_ctx *ctx, size_t bytes, void **dptr);
int cp_free(cp_ctx *ctx, void *dptr);
int cp_memcpy_h2d(cp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int cp_memcpy_d2h(cp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes); /* syncs */
int cp_memset(cp_ctx *ctx, void *dst_dev, int value, size_t bytes);

/* ---- a1: sampled-point im2col --------------------------------------------------- */
/* Replaces the window copy of Net.extract_XY (lib/net.py:629-657, non-gw1 branch)
 * + the reshape/rollaxis of lib/net.py:1702 + the VGG ReLU of lib/net.py:1720, for ONE
 * batch: fmap DEVICE [B,C,H,W] f32 (bottom blob, unpadded), xs/ys HOST int32[P] sampled
 * TOP coordinates.  Writes rows [row0, row0+P*B) of X_out DEVICE [*, C, k, k] f32 in the
 * reference's row order [point][image]; zero padding `pad`, window origin x*stride-pad. */
int cp_patch_gather(cp_ctx *ctx, const float *fmap, int B, int C, int H, int W, const int32_t *xs,
                    const int32_t *ys, int P, int k, int pad, int stride, int relu, float *X_out,
                    int64_t row0);
/* The same for ALL nb batches of a layer in one launch (the loop over batches of lib/net.py:622-657):
 * fmap DEVICE [nb,B,C,H,W] f32, xs/ys HOST int32[nb*P] batch-major; writes the nb*P*B rows of X_out. */
int cp_patch_gather_batches(cp_ctx *ctx, const float *fmap, int nb, int B, int C, int H, int W,
                            const int32_t *xs, const int32_t *ys, int P, int k, int pad, int stride, int relu,
                            float *X_out);

/* ---- a2: target assembly -------------------------------------------------------- */
/* Y = feats - bias (+ resY): lib/net.py:1707,1716-1722.  feats DEVICE [N,n] f32,
 * bias DEVICE [n] f32, resY DEVICE [N,n] f64 or NULL, Y DEVICE [N,n] f64. */
int cp_assemble_y(cp_ctx *ctx, const float *feats, const float *bias, const double *resY, int64_t N,
                  int n, double *Y);

/* ---- a3 (first half): LASSO operands -------------------------------------------- */
/* Replaces lib/decompose.py:428-437 (Z = matmul(reX, reW2).reshape(c,-1).T; reY) and
 * the centring + Gram that sklearn's Lasso.fit needs (_base.py:108-205): with
 * Z[(s,j),i] = sum_t X[samples[s],i,t] W2[j,i,t], zc = Z - colmean, yc = y - mean:
 *   Q = zc^T zc  DEVICE [c,c] f64,  q = zc^T yc DEVICE [c] f64,
 *   stats DEVICE f64[4] = {yc^T yc, mean(y), M = S*n, 0}.
 * Z itself is never returned.  X DEVICE [N,c,kk] (x_dtype), W2 DEVICE [n,c,kk]
 * (w_dtype), Y DEVICE [N,n] f64, samples HOST int64[S] (rows may repeat). */
int cp_lasso_gram(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2,
                  int w_dtype, int n, const double *Y, const int64_t *samples, int S, double *Q,
                  double *q, double *stats);

/* ---- a4: one LASSO fit = sklearn enet_coordinate_descent_gram -------------------- */
typedef struct cp_cd_result {
    double gap;        /* duality gap at exit (unscaled, as _cd_fast returns it) */
    double tol_scaled; /* tol * yc^T yc */
    int32_t n_iter;    /* epochs run (sklearn's n_iter_) */
    int32_t nnz;       /* count of w != 0 (the reference's sum(idxs), decompose.py:465) */
    /* Tie sentinels: how close the fit's discrete decisions came to flipping.  The reference runs scikit-learn's DATA
     * form of the recurrence; any Gram form agrees with it to rounding only, so a decision taken within a few ulp of its
     * threshold is one the reference may have taken the other way.  -1 = not tracked (kernel forms
     * of cd_gram.hip; the team kernels of cd_team.hip track both). */
    double edge_margin; /* min over the coordinates, each at its LAST update of the fit, of | |q_i - H_i| - l1 | / l1: distance
                           of a coefficient from the edge of its dead zone (zero <-> non-zero), relative to l1 */
    double gap_margin;  /* min over the fit's duality-gap tests of |gap - tol_scaled| / tol_scaled (stop <-> one more epoch) */
} cp_cd_result;

#define CP_CD_RECIPROCAL 1 /* multiply by 1/(Qii+l2) instead of dividing (<=1 ulp/step) */
#define CP_CD_DELTA 2      /* one axpy H += (w_new - w_old) Q[ii] instead of sklearn's two (rounding-level) */
/* cp_prune_layer / cp_prune_layers only (same flags word): compute the normal equations over ALL c channels on the device's
 * side stream while the single-workgroup alpha search runs; the refit then gathers the kept rows / columns instead of
 * running its two N-sized products after the search.  (c / kept)^2 times the Gram flops, off the critical path. */
#define CP_REFIT_PRECOMPUTE 4
/* with CP_REFIT_PRECOMPUTE, when rank >= 0.8 c: also factor that full Gram and forward-substitute the right-hand side
 * during the search; the refit is then a constrained solve with the full factor (dropped coefficients forced to zero)
 * -- no factorisation of the kept sub-matrix after the search.  Falls back to it when a pivot of the full Gram fails. */
#define CP_REFIT_PREFACTOR 8

/* Replaces Lasso.fit as called by solve() (lib/decompose.py:453-466):
 * sklearn/_cd_fast.pyx:564-737 with random coordinate order from our_rand_r
 * (sklearn/utils/_random.pxd:20-35) seeded by `seed` (the value the host drew with
 * rng.randint(0, 2147483647), _cd_fast.pyx:626).  Q DEVICE [c,c] (row stride ldq), q
 * DEVICE [c], stats DEVICE (stats[0] = yc^T yc), w DEVICE [c] in/out (warm start),
 * result HOST.  l1_reg = alpha*M, l2_reg = 0 for Lasso. */
int cp_enet_cd_gram(cp_ctx *ctx, const double *Q, int ldq, const double *q, const double *stats, int c,
                    double l1_reg, double l2_reg, uint32_t seed, int max_iter, double tol, int flags,
                    double *w, cp_cd_result *result);

/* Which kernel form cp_enet_cd_gram / cp_lasso_alpha_search run for c channels and these flags (no GPU needed):
 * 0 = one wavefront, 1 = chain wave + assist / keeper wave, 2 = team (chain wave + up to six keeper waves, one workgroup),
 * 3 = multi-CU team (512 < c <= 2048: the keepers in 2-4 further workgroups, one CU each); -1 = c not supported.
 * All forms produce the same bits; the choice follows c, the flags and the CP_CD_* environment switches. */
#define CP_CD_FORM_WAVE 0
#define CP_CD_FORM_DUO 1
#define CP_CD_FORM_TEAM 2
#define CP_CD_FORM_MULTI 3
int cp_cd_kernel_form(int c, int flags);

/* The whole alpha search of lib/decompose.py:490-525 in ONE launch (bracket doubling
 * then bisection; acceptance lbound <= nnz <= rbound), warm-starting every fit from
 * the previous one exactly as Lasso(warm_start=True) does.  seeds HOST uint32[max_fits]
 * are the values the host pre-drew from the RNG; *fits_used tells it how many draws
 * the reference would have consumed (the host rewinds its RNG accordingly).
 * M = S*n (alpha -> l1_reg = alpha*M).  fit_log HOST [max_fits] (alpha per fit in
 * fit_alpha HOST double[max_fits]) may be NULL.  w DEVICE [c] out (start = 0). */
int cp_lasso_alpha_search(cp_ctx *ctx, const double *Q, int ldq, const double *q, const double *stats,
                          int c, double M, double alpha_right0, double rank, double lbound,
                          double rbound, const uint32_t *seeds, int max_fits, int max_iter, double tol,
                          int flags, double *w, int *fits_used, double *alpha_out,
                          cp_cd_result *fit_log, double *fit_alpha);

/* ---- a5: least-squares refit ---------------------------------------------------- */
typedef struct cp_refit_info {
    int32_t p;        /* columns = kept channels * kk */
    int32_t rank;     /* numerical rank: p on the Cholesky paths, gelsd's rank (sigma_i > max(N,p) eps sigma_max) on path 3,
                         -1 on path 1 (not determined) */
    int32_t fallback; /* 0 = Cholesky of the normal equations (every pivot above 1e-6 of its diagonal);
                         2 = shifted-Cholesky-QR preconditioning, then Cholesky (ill-conditioned, full column rank);
                         3 = preconditioning + two one-sided Jacobi decompositions: minimum-norm solution with gelsd's
                             cut-off (rank-deficient: copies of channels, N <= p, ...);
                         1 = iterated Tikhonov (ridge > 0, and the row-sharded tail, which never sees the rows) */
    int32_t reserved;
} cp_refit_info;

/* Replaces fc_kernel(X[:,idxs].reshape(N,-1), Y) (lib/decompose.py:622, 636-669):
 * LinearRegression(fit_intercept=True) -> centre, minimum-norm least squares
 * (scipy gelsd, cut-off max(N,p)*eps), intercept = ybar - xbar.coef^T; ridge > 0 selects
 * the Ridge branch (decompose.py:662-663).  As accurate as gelsd (error ~ cond * eps) whatever the
 * conditioning: see cp_refit_info.fallback.  X DEVICE [N,c,kk] (x_dtype), mask HOST
 * uint8[c] (non-zero = keep), Y DEVICE [N,n] f64.  Outputs DEVICE: W_out [n, p] f64
 * (p = kept*kk, caller reshapes to [n, kept, k, k]), b_out [n] f64; info HOST. */
int cp_lstsq_refit(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk,
                   const uint8_t *mask, const double *Y, int n, double ridge, double *W_out,
                   double *b_out, cp_refit_info *info);

/* fc_kernel (lib/decompose.py:636-669) with the ROWS of X / Y spread over several ranks (one process per GPU;
 * SURVEY.md section 8e "shard N across GPUs for the Gram builds, one all-reduce each").  The library does the
 * per-rank arithmetic, the caller runs the two sum all-reduces (RCCL via torch.distributed) on buffers it owns:
 *
 *   cp_refit_shard_layout(kept, kk, n, &sums_elems, &gram_elems)      sizes (f64 elements) of the two buffers
 *   cp_refit_shard_sums (.., X_local, N_local, .., sums)              column sums of this rank's rows
 *       all-reduce(sum) sums[sums_elems]
 *   cp_refit_shard_gram (.., X_local, N_local, .., N_total, sums, gram)
 *       centres this rank's rows with the GLOBAL means (what LinearRegression's _preprocess_data does,
 *       sklearn/linear_model/_base.py:108-205) and writes Xs^T Xs | Xs^T Yc of them
 *       all-reduce(sum) gram[gram_elems]
 *   cp_refit_shard_solve(.., N_total, ridge, sums, gram, W_out, b_out, info)
 *       the factor / substitute / intercept tail of cp_lstsq_refit, run on every rank on identical inputs.
 *
 * X_local DEVICE [N_local, c, kk] (x_dtype), Y_local DEVICE [N_local, n] f64, mask HOST uint8[c], sums / gram DEVICE
 * f64 (written by _sums / _gram, read-only afterwards), W_out / b_out DEVICE as cp_lstsq_refit.  Every call returns
 * with its stream drained, so the collective may run on any other stream. */
int cp_refit_shard_layout(int kept, int kk, int n, int64_t *sums_elems, int64_t *gram_elems);
int cp_refit_shard_sums(cp_ctx *ctx, const void *X, int x_dtype, int64_t N_local, int c, int kk, const uint8_t *mask,
                        const double *Y, int n, double *sums);
int cp_refit_shard_gram(cp_ctx *ctx, const void *X, int x_dtype, int64_t N_local, int c, int kk, const uint8_t *mask,
                        const double *Y, int n, int64_t N_total, const double *sums, double *gram);
int cp_refit_shard_solve(cp_ctx *ctx, int kept, int kk, int n, int64_t N_total, double ridge, const double *sums,
                         const double *gram, double *W_out, double *b_out, cp_refit_info *info);

/* Replaces nonlinear_fc(X[:, idxs].reshape(N, -1), Y) (lib/decompose.py:671-685; the dcfgs.nonlinear_fc
 * branch of dictionary(), decompose.py:615-617): Z = relu(Y), U = Y, then for every stage s (reference: iters =
 * {30, 20}, lambdas = {0.1, 1}) iters[s] times  reg = fc_kernel(X, U);  U = solve_relu(reg.predict(X), Z, lambdas[s])
 * (decompose.py:51-59).  X is constant over the 50 regressions: it is centred, its Gram factorised once, each
 * iteration is two GEMMs, one substitution launch and one element-wise pass.  Returns the coefficients /
 * intercept of the LAST regression like the reference.  Arguments as cp_lstsq_refit (outputs DEVICE); needs
 * N - 1 >= p and a positive definite Gram (CP_ERR_UNSUPPORTED / CP_ERR_NUMERIC otherwise). */
int cp_nonlinear_fc(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const uint8_t *mask,
                    const double *Y, int n, const int *iters, const double *lambdas, int n_stage,
                    double *W_out, double *b_out, cp_refit_info *info);

/* ---- f1: VH_decompose (spatial decomposition, lib/decompose.py:85-146) ---------------------------- */
/* Replaces svd(VH) = scipy.linalg.svd(x, full_matrices=False, lapack_driver='gesvd') (decompose.py:45-47, 100)
 * together with the truncation of decompose.py:105-112: one-sided Jacobi on the rows of M DEVICE [m, n]
 * (row-major f64, m <= n, untouched).  Outputs DEVICE: sigma [r] descending, Vt [r, m] (row k = V[:, k], the
 * k-th left singular vector; sign arbitrary as with LAPACK), SH [r, n] = diag(sigma) H[:r].  *sweeps HOST. */
int cp_svd_rows(cp_ctx *ctx, const double *M, int m, int n, int r, double *sigma, double *Vt, double *SH,
                int *sweeps);
/* The same for a matrix whose numerical rank is at most r -- svd(T) at the end of ITQ_decompose (decompose.py:249-252: T is
 * the rank-`rank` map of the last alternation): rows that fall below 1e-13 of the largest row norm while the sweeps run
 * are left alone instead of being rotated against each other (they carry rounding noise only; half the sweeps). */
int cp_svd_rows_lowrank(cp_ctx *ctx, const double *M, int m, int n, int r, double *sigma, double *Vt, double *SH,
                        int *sweeps);
/* Xv[s, r*w + wi] = sum_{ci,hi} X[s,ci,hi,wi] V[(ci,hi), r]: np.tensordot(X, V, [[1,2],[1,2]]) + the transpose and
 * reshape of decompose.py:131-136.  X DEVICE [N,c,h,w] (x_dtype), Vt DEVICE [rank, c*h], Xv DEVICE [N, rank*w]. */
int cp_vh_project(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int h, int w, const double *Vt,
                  int rank, double *Xv);
/* C[m,n] = A^T B, A DEVICE [k,m], B DEVICE [k,n], C DEVICE [m,n] (row-major f64, any sizes): the small
 * reconstructions V.dot(H) of decompose.py:114, 139. */
int cp_matmul_tn(cp_ctx *ctx, const double *A, const double *B, int m, int n, int k, double *C);

/* ---- f2: ITQ_decompose (channel decomposition, lib/decompose.py:163-319) ---------------------------- */
/* The 30 + 20 alternations of decompose.py:163-246 on the device: G = Y - mean, PGGt = pinv(G^T G, cond) G^T,
 * then per iteration  X = G (PGGt UU);  T = rank-truncated SVD of X (cp_svd_rows on X^T);  T = PGGt T;
 * U = solve_relu(G T + U_mean, relu(gt), lambda);  U_mean = mean(U), UU = U - U_mean.  feature / gt_feature
 * DEVICE [N, n] f64; outputs DEVICE: T [n, n] (decompose.py:226), ymean [n], umean [n] (last U_mean).
 * pinv_cond = 1e-6 in the reference (scipy.linalg.pinv(x, 1e-6), decompose.py:148-151). */
int cp_itq_iterate(cp_ctx *ctx, const double *feature, const double *gt_feature, int64_t N, int n, int rank,
                   const int *iters, const double *lambdas, int n_stage, double pinv_cond, double *T_out,
                   double *ymean_out, double *umean_out);

/* ---- a3: one whole dictionary() call -------------------------------------------- */
#define CP_MAX_FITS 64
typedef struct cp_prune_result {
    int32_t fits_used;  /* LASSO fits consumed (= RNG draws the host replays); 0 when rank >= c;
                           -1: the search did not settle within max_fits (nothing else is valid) */
    int32_t nnz;        /* kept channels = sum(mask) */
    int32_t p;          /* nnz * kk: columns of W_out */
    int32_t refit_rank; /* cp_refit_info.rank */
    int32_t fallback;   /* cp_refit_info.fallback */
    int32_t uploaded;   /* cp_prune_layer_h2d: 1 once the copies of X_host / Y_host into X_dev / Y_dev were enqueued
                           (they are complete when the call returns, whatever it returns); 0: an error return came
                           before that and X_dev / Y_dev hold nothing */
    double alpha;       /* alpha of the accepted fit (decompose.py:525) */
    cp_cd_result fit_log[CP_MAX_FITS];
    double fit_alpha[CP_MAX_FITS];
} cp_prune_result;

/* Replaces the device work of one dictionary() call (lib/decompose.py:425-437, 453-466,
 * 487-525, 622-623) in one foreign call: cp_lasso_gram -> cp_lasso_alpha_search -> mask = (w != 0)
 * -> cp_lstsq_refit -> copies back.  X / W2 / Y DEVICE as for cp_lasso_gram; samples HOST
 * int64[S]; seeds HOST uint32[max_fits] pre-drawn by the caller, who rewinds its RNG and
 * re-draws res->fits_used of them; lbound/rbound as decompose.py:493-501 computes them.
 * rank >= c skips the LASSO (decompose.py:487-488).  Outputs HOST: mask_out uint8[c],
 * W_out f64 [n, p] (capacity n*c*kk), b_out f64 [n], res.  Synchronises the stream.
 * W_out == b_out == NULL: the results are not copied; borrow them with cp_result_host. */
int cp_prune_layer(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2,
                   int w_dtype, int n, const double *Y, const int64_t *samples, int S,
                   double alpha_right0, double rank, double lbound, double rbound,
                   const uint32_t *seeds, int max_fits, int max_iter, double tol, int flags,
                   double ridge, uint8_t *mask_out, double *W_out, double *b_out,
                   cp_prune_result *res);
/* The same call for operands that still sit in the caller's host arrays -- what the reference's dictionary(X, W2, Y, ...)
 * (lib/decompose.py:386) is handed: X_dev / Y_dev are device buffers of the right size that the call FILLS from X_host
 * [N, c, kk] (x_dtype) / Y_host [N, n] (float64).  Only the S sampled rows are uploaded before the LASSO operands and the
 * alpha search start; the rest streams in on a second stream while the search runs, and the refit is ordered after it.
 * Results are bit-identical to cp_prune_layer on resident operands.  W2: device, as in cp_prune_layer. */
int cp_prune_layer_h2d(cp_ctx *ctx, void *X_dev, const void *X_host, int x_dtype, int64_t N, int c, int kk, const void *W2,
                       int w_dtype, int n, double *Y_dev, const double *Y_host, const int64_t *samples, int S,
                       double alpha_right0, double rank, double lbound, double rbound, const uint32_t *seeds, int max_fits,
                       int max_iter, double tol, int flags, double ridge, uint8_t *mask_out, double *W_out, double *b_out,
                       cp_prune_result *res);

/* HOST pointers to the b [n] and W [n, p] of the last cp_prune_layer (or cp_prune_layers job) on this context, in
 * the context's page-locked result block, where the last refit kernel wrote them: no copy, DMA-able, valid until the
 * next call on the context.  No reference counterpart (dictionary() returns fresh arrays, lib/decompose.py:634). */
int cp_result_host(cp_ctx *ctx, const double **b, const double **W, int *n, int *p);

/* The same for several independent layers at once (e.g. the equal-shaped layers of a ResNet stage, or one layer of
 * several networks): jobs[i] is exactly the argument list of cp_prune_layer.  All jobs must have the same channel
 * count c; ctxs[i] are DISTINCT contexts on ONE stream and device (a context and its cp_ctx_create_sibling()s, or
 * contexts bound to a caller-owned stream with cp_ctx_set_stream) -- each keeps its own workspace and pinned result
 * block.  No reference counterpart: the reference prunes one layer per dictionary() call (lib/net.py:1443).  A stream runs one kernel at a time and most of a layer's time is its
 * single-workgroup alpha search, so the searches of the batch are the workgroups of one launch; the host waits twice
 * per call.  results[i] as cp_prune_layer's (fits_used == -1: that layer's search did not settle, nothing else of
 * results[i] / its outputs is valid).  At most CP_MAX_JOBS jobs per call. */
#define CP_MAX_JOBS 16
typedef struct cp_prune_job {
    const void *X;
    int32_t x_dtype;
    int32_t c;
    int64_t N;
    int32_t kk;
    int32_t w_dtype;
    const void *W2;
    int32_t n;
    int32_t S;
    const double *Y;
    const int64_t *samples;
    double alpha_right0, rank, lbound, rbound;
    const uint32_t *seeds;
    int32_t max_fits, max_iter;
    double tol;
    int32_t flags;
    int32_t reserved;
    double ridge;
    uint8_t *mask_out;
    double *W_out;
    double *b_out;
} cp_prune_job;
int cp_prune_layers(int n_jobs, cp_ctx *const *ctxs, const cp_prune_job *jobs, cp_prune_result *results);

/* ---- micro-benchmarks used by bench.py for roofline denominators ---------------- */
/* Sustained v_mfma_f64_16x16x4_f64 rate (TFLOP/s) and float4-copy HBM bandwidth (GB/s). */
int cp_probe_mfma_f64(cp_ctx *ctx, double *tflops);
/* The same probe with every wave stamping the shader-clock counter (s_memtime) and the constant 100 MHz counter
 * (s_memrealtime) around its loop: tflops as above, ghz = the shader clock the chip held during the launch (median wave),
 * cycles_per_mfma = shader cycles per v_mfma_f64_16x16x4_f64 and SIMD at that rate (64 would be the nominal 78.6 TFLOP/s). */
int cp_probe_mfma_f64_clock(cp_ctx *ctx, double *tflops, double *ghz, double *cycles_per_mfma);
int cp_probe_hbm_copy(cp_ctx *ctx, size_t bytes, double *gbps);

/* Test hook (host only, no GPU needed): how a launch of the f64 "TN" GEMM behind the Gram builds (C[M,N] = A^T B, tri as in
 * the library: 0 general, 1 lower tiles + mirror, 2 upper tiles) divides its 128 x 128 tiles over its workgroups on a chip of
 * cu_count CUs -- whole tiles, the tail tiles split along K (added by the last arrival), uniform split-K.  Computed by the same
 * functions the launch and the kernel use.  plan[9] = {n_tiles, tiles_n, small, planes, n_full, n_split, s, kchunk, units};
 * units (or NULL): 5 int32 per workgroup = {tile row, tile column, chunk, chunks of the tile, idle}.  Returns the number of
 * workgroups, or -CP_ERR_ARG. */
int cp_debug_gemm_units(int cu_count, int M, int N, int K, int tri, int32_t *plan, int32_t *units, int max_units);

/* Per-stage device timings (ms, HIP events on the ctx stream) of the most recent
 * cp_lasso_gram / cp_lstsq_refit call; names in cp_stage_name().  Used by bench.py. */
#define CP_MAX_STAGES 32
int cp_last_stage_times(cp_ctx *ctx, int *count, float *ms /* [CP_MAX_STAGES] */);
const char *cp_stage_name(cp_ctx *ctx, int index);
int cp_enable_stage_timing(cp_ctx *ctx, int on); /* 0 off, 1 every stage, 2 only "refit_gram_gemm" (two events per
                                                   * call: each event is one more packet in the stream) */
/* A common clock for the stage brackets of SEVERAL contexts (the layers of a job run on their own streams): cp_stage_epoch
 * records a reference event on ctx's stream; cp_last_stage_spans is cp_last_stage_times that also returns, per stage, when
 * its bracket BEGAN in ms after the epoch of `epoch_of` (another context of the same device, or ctx itself): the wall window
 * that concurrent brackets span = max(begin + ms) - min(begin).  Used by bench.py for roofline.chip_level. */
int cp_stage_epoch(cp_ctx *ctx);
int cp_last_stage_spans(cp_ctx *ctx, cp_ctx *epoch_of, int *count, float *ms /* [CP_MAX_STAGES] */,
                        float *begin_ms /* [CP_MAX_STAGES] */);

#ifdef __cplusplus
}
#endif
#endif /* CPMI355_H */
