// Internal declarations shared by the HIP translation units of libcpmi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cpmi355.h"

// The cross-workgroup hand-offs of chol_step.hip and gemm_f64.hip (relaxed agent-scope `sc1` stores, each wave's own
// `s_waitcnt vmcnt(0)`, a relaxed flag, `sc1` loads on the consumer side, no acquire / release fence) lean on how gfx942 /
// gfx950 treat sc1 accesses: written through to, and read from, the memory side of the XCD L2s.  That is outside what the
// LLVM AMDGPU memory model promises in general, so the device code refuses to build for anything else;
// -DCP_HANDOFF_FENCES=1 puts agent-scope release / acquire fences around the same hand-offs (tests/: the parity suite runs
// against either build).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "libcpmi355 device code is written for gfx950 (MI355X); its fence-free sc1 hand-offs are not valid on other targets"
#endif
#ifndef CP_HANDOFF_FENCES
#define CP_HANDOFF_FENCES 0
#endif
// release side: after the data stores, before the flag is raised; acquire side: after the flag was seen, before the data loads
#if CP_HANDOFF_FENCES
#define CP_HANDOFF_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define CP_HANDOFF_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define CP_HANDOFF_RELEASE() ((void)0)
#define CP_HANDOFF_ACQUIRE() ((void)0)
#endif

// what a deferred refit (cp_prune_layers) leaves behind: everything its factorisation and substitution launches need,
// so that the batch can run them as ONE launch each for all of its layers
struct cp_refit_deferred {
    double *G, *U, *Lt, *TI, *TIT, *dg0, *gmax, *Rm;
    double *part;   // scratch of the tiled lay-out kernel ((p_pad / 32) * n_pad doubles)
    int *info;
    int p, p_pad, nblk, n, n_pad;
    const double *xmean, *ymean;
    double *W_out, *b_out, *W_host, *b_host;
    int *info_host;
};

// The normal equations of a layer over ALL its c channels, computed on the device's shared side stream WHILE the layer's
// (single-workgroup, latency-bound) alpha search runs: once the mask is known the refit only gathers the kept rows /
// columns (cp_refit_precompute_enqueue, refit.hip).
struct cp_ctx;
struct cp_precompute {
    bool ready = false;            // set by enqueue, consumed (one shot) by the refit cp_prune_layer(s) issues next
    bool armed = false;            // only while cp_prune_layer(s) is inside that refit: no other call may consume `ready`
    const void *X = nullptr;
    const double *Y = nullptr;
    int64_t N = 0;
    int c = 0, kk = 0, n = 0, x_dtype = 0, P = 0, P_pad = 0, n_pad = 0;
    char *buf = nullptr;           // persistent: xmean [P_pad] | ymean [n_pad] | G [P_pad^2] | R [P_pad n_pad]
    size_t buf_bytes = 0;
    double *xmean = nullptr, *ymean = nullptr, *G = nullptr, *R = nullptr;
    hipEvent_t done = nullptr;
    cp_ctx *worker = nullptr;      // own arena, bound to the side stream
    // The factorisation of the FULL Gram and the forward-substituted right-hand side, also computed during the search
    // (rank hint >= 0.8 c): the refit then solves the kept-channel problem as an equality-constrained one with this
    // factor (refit.hip: refit_from_full_factor) instead of factoring the kept sub-matrix after the search.
    bool factored = false;
    int nblk = 0;
    char *fbuf = nullptr;          // persistent: Gw | U | Lt (P_pad^2 each) | TI | TIT | F [P_pad n_pad] | dg0 | gmax | info
    size_t fbuf_bytes = 0;
    double *Gw = nullptr, *U = nullptr, *Lt = nullptr, *TI = nullptr, *TIT = nullptr, *F = nullptr, *dg0 = nullptr, *gmax = nullptr;
    int *finfo = nullptr;
    hipStream_t chain_stream = nullptr;   // this context's own stream for the (latency-bound) factorisation chain
    hipEvent_t gram_done = nullptr;
};

struct cp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // Grow-only scratch arena: one hipMalloc, re-grown (after a stream sync) when a call
    // needs more.  All intermediates of one call are carved from it.
    char *arena = nullptr;
    size_t arena_bytes = 0;
    size_t arena_used = 0;
    // per-layer outputs of cp_prune_layer (Q, q, stats, w, W, b): persistent across its sub-calls
    char *layer_ws = nullptr;
    size_t layer_ws_bytes = 0;
    int result_n = 0, result_p = 0;    // shape of the (b, W) the last cp_prune_layer left in the pinned block
    double host_ms[4] = {0, 0, 0, 0};  // host wall time of the phases of the last cp_prune_layer
    double wait_ms = 0;                // of which: blocked in cp_stream_wait (running total)
    // pinned host staging for small D2H results
    char *pinned = nullptr;
    size_t pinned_bytes = 0;
    const double *pinned_w = nullptr;  // host copy of w written by the last alpha-search kernel
    char err[512] = {0};
    // stage timing: a list of (name, event); name == nullptr marks the start of a call
    bool timing = false;
    bool timing_gram_only = false;  // cp_enable_stage_timing(ctx, 2)
    int n_marks = 0;
    hipEvent_t ev[2 * CP_MAX_STAGES] = {};
    hipEvent_t ev_epoch = nullptr;    // cp_stage_epoch: the common clock of several contexts' stage brackets
    const char *mark_names[2 * CP_MAX_STAGES] = {};
    int n_stages = 0;
    const char *stage_names[CP_MAX_STAGES] = {};
    float stage_ms[CP_MAX_STAGES] = {};
    const char *gemm_mark = nullptr;  // if set, cp_gemm_tn_f64 marks this stage right after its main kernel
    int gemm_tag = 0;                 // selects a distinctly named instantiation of the GEMM kernel
    int cu_count = 256;
    int *gemm_cnt = nullptr;          // arrival counters of the split tiles of cp_gemm_tn_f64 (a ring of regions, zero between launches)
    int gemm_cnt_next = 0;
    bool defer_refit_wait = false;    // cp_prune_layers: enqueue the refit, the caller waits once for the whole batch
    bool refit_pending = false;       // set by a deferred refit: factor + solve still to be launched by the batch
    cp_refit_deferred deferred = {};
    cp_precompute pre;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   // hand-offs to / from the shared CU-masked stream of the long GEMMs
    int itq_sweeps = 0;               // Jacobi sweeps of the last cp_itq_iterate (all alternations)
    int itq_ns_steps = 0, itq_sign_alternations = 0;   // Newton-Schulz steps / alternations that took the sign-function route
    char *cd_box = nullptr;           // mailboxes of the multi-CU coordinate-descent team (cd_team.hip), grow-only
    size_t cd_box_bytes = 0;
    char *stage = nullptr;             // page-locked staging of the sampled rows (cp_prune_layer_h2d), grow-only
    size_t stage_bytes = 0;
    hipEvent_t ev_upload = nullptr;    // the side-stream uploads of cp_prune_layer_h2d have landed
    bool last_cd_was_team = false;    // which kernel family the last coordinate-descent launch of THIS context ran (debug counters)
    int cd_fallbacks = 0;             // searches / fits re-run on the one-workgroup team after a hand-off time-out of the multi-CU team
    int chol_test_fail_flag_waits = 0;   // cp_debug_chol_fail_flag_wait: that many factorisations run with a spin limit of 0 (tests)
    bool cd_test_fail_multi = false;  // cp_debug_cd_fail_multi: the next multi-CU launches give up at once (tests of that fallback)
    bool last_xty_fused = false;      // the last refit of this context formed G and X^T Y in one launch (cp_gemm_gram_xty)
};

// process-wide experiment switches (cp_debug_knob; cp_ctx.hip): plain ints, 0 by default, read when work is enqueued
enum { CP_KNOB_SPLIT_XTY = 0, CP_KNOB_CHOL_PHASES = 1, CP_KNOB_CHOL_WG = 2, CP_KNOB_COUNT = 8 };
int cp_knob(int id);

int cp_set_error(cp_ctx *ctx, int code, const char *fmt, ...);

#define CP_HIP(ctx, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return cp_set_error((ctx), CP_ERR_HIP, "%s failed: %s (%s:%d)", #call,             \
                                hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

#define CP_TRY(expr)               \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != CP_OK) return rc_; \
    } while (0)

#define CP_LAUNCH_CHECK(ctx) CP_HIP(ctx, hipGetLastError())

hipStream_t cp_side_stream(cp_ctx *ctx);   // the device's shared stream for work that overlaps a context's own chain (never null)
// A pending precompute that nobody will consume (error exit, an unrelated refit, new contents in its buffers): wait for the
// side / chain stream work that still reads X / Y, then forget it.
void cp_precompute_void(cp_ctx *ctx);
int cp_refit_precompute_enqueue(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const double *Y, int n,
                                double rank_hint = 0.0, bool fork_recorded = false);
void cp_precompute_release(cp_ctx *ctx);
// hipStreamSynchronize(ctx->stream), with the time spent blocked added to ctx->wait_ms
hipError_t cp_stream_wait(cp_ctx *ctx);

// arena ------------------------------------------------------------------------------
int cp_arena_reserve(cp_ctx *ctx, size_t bytes);  // ensure capacity (may sync + realloc); resets used=0
void *cp_arena_take(cp_ctx *ctx, size_t bytes);   // 256-B aligned carve; nullptr if exhausted
template <typename T>
static inline T *cp_arena_take_t(cp_ctx *ctx, size_t count) {
    return reinterpret_cast<T *>(cp_arena_take(ctx, count * sizeof(T)));
}
static inline size_t cp_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int cp_pinned_reserve(cp_ctx *ctx, size_t bytes);

// stage timing -----------------------------------------------------------------------
void cp_stage_begin(cp_ctx *ctx);                   // start of a top-level call
void cp_stage_mark(cp_ctx *ctx, const char *name);  // closes the stage that just ran
void cp_stage_finish(cp_ctx *ctx);                  // kept for symmetry (times are resolved lazily)

// f64 GEMM (gemm_f64.hip) --------------------------------------------------------------
// C[M,N] = alpha * sum_k A[k,m] * B[k,n] + beta * C   (both operands k-major, "TN").
// Requirements: M % 128 == 0, N % 128 == 0, K % 16 == 0, lda/ldb/ldc even, 16-B aligned
// bases.  tri: 0 general, 1 lower tiles only (tile_n <= tile_m) + mirror to the upper
// part (A and B must then describe the same matrix), 2 upper tiles only (no mirror).
// Deterministic: split-K partials are reduced in a fixed order.
enum { CP_TRI_NONE = 0, CP_TRI_LOWER_MIRROR = 1, CP_TRI_UPPER = 2 };
// kernel-name tags (ctx->gemm_tag) so that a profiler lists the big contractions separately
enum { CP_GEMM_GENERIC = 0, CP_GEMM_LASSO_GRAM = 1, CP_GEMM_REFIT_GRAM = 2, CP_GEMM_REFIT_XTY = 3 };
int cp_gemm_tn_f64(cp_ctx *ctx, int M, int N, int K, double alpha, const double *A, int lda,
                   const double *B, int ldb, double beta, double *C, int ldc, int tri);
size_t cp_gemm_tn_workspace(const cp_ctx *ctx, int M, int N, int K, int tri);
// G = X^T X (both triangles) and R = X^T Y in one launch (gemm_f64.hip); *fused == false: nothing was launched (plan with
// reduction planes), run the two products separately
int cp_gemm_gram_xty(cp_ctx *ctx, int p_pad, int n_pad, int K, const double *X, int ldx, const double *Y, int ldy, double *G,
                     int ldg, double *R, int ldr, bool *fused);
size_t cp_gemm_gram_xty_workspace(const cp_ctx *ctx, int p_pad, int n_pad, int K);
// two products of the same shape (different operands / K), one launch when neither needs split-K
int cp_gemm_tn_f64_pair(cp_ctx *ctx, int M, int N, double alpha, int K1, const double *A1, const double *B1, double *C1,
                        int K2, const double *A2, const double *B2, double *C2, int lda, int ldb, int ldc, int tri);

// cp_lstsq_refit with optional host-visible outputs: b (n doubles) then W (n x p) at ctx->pinned + 64
// batched alpha search (cd_gram.hip): one launch on ctxs[0]->stream, one workgroup per job; results land in each
// context's pinned block and are read by cp_alpha_search_collect after the stream has been waited on
struct cp_search_job {
    const double *Q;
    int ldq;
    const double *q, *stats;
    int c;
    double M, alpha_right0, rank, lbound, rbound;
    const uint32_t *seeds;
    int max_fits, max_iter;
    double tol;
    int flags;
    double *w;
};
// chol_step.hip: blocked Cholesky, one launch per 128-column step
// info of a factorisation (zeroed by the caller: k_diag_prepare / k_add_diag_scaled): [0] first failed pivot + 1 (also NaN;
// 0x7fffffff: a bounded wait ran out), [1 + b] block columns of the operator of block b published; from cp_chol_ctl_offset on,
// the control block of the persistent form: [0] task counter, [1] stop word, [8 + i (nblk + 32) + x] version word of tile (i, x)
constexpr int cp_chol_ctl_offset(int nblk) { return 8 + 2 * nblk; }
constexpr int cp_chol_info_count(int nblk) { return cp_chol_ctl_offset(nblk) + 8 + nblk * (nblk + 32); }
int cp_chol_factor_steps(cp_ctx *ctx, double *G, double *U, double *Lt, int ld, int nblk, const double *dg0,
                         double piv_tol, double *TI, double *TIT, int *info, double *R = nullptr, int n_pad = 0);
int cp_alpha_search_enqueue_batch(cp_ctx *const *ctxs, int n_jobs, const cp_search_job *jobs, bool allow_multi = true);
// timed_out (optional): a fit of the search reported a hand-off time-out of its team (n_iter = -1): run it again with
// allow_multi = false (the one-workgroup team, bit-identical)
int cp_alpha_search_collect(cp_ctx *ctx, int c, int max_fits, int *fits_used, double *alpha_out, cp_cd_result *fit_log,
                            double *fit_alpha, bool *timed_out = nullptr);
// factor + substitute every pending deferred refit of the batch: two launches on ctxs[0]->stream (refit.hip)
int cp_refit_batch_factor_solve(cp_ctx *const *ctxs, int n_ctx);
int cp_lstsq_refit_impl(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const uint8_t *mask,
                        const double *Y, int n, double ridge, double *W_out, double *b_out, cp_refit_info *info,
                        bool host_out);

// one-sided Jacobi SVD (svd_jacobi.hip) with caller-provided scratch -----------------------------------
// Scratch of one decomposition (device): Wk [me, n], R [me, me], sig [me], rotated [16 ints], order [me ints]
static inline int cp_svd_me(int m) { return (m + 15) / 16 * 16; }   // rows the Jacobi kernels work on (zero rows pad)
struct SvdScratch {
    double *Wk, *R, *sig;
    int *rotated, *order;
    // host copies left by cp_svd_rows_core: the r-th and (r + 1)-th singular value (0 if there is none) and the sum of all
    double sigma_r = 0, sigma_next = 0, sigma_sum = 0;
    static size_t bytes(int m, int n) {
        const size_t me = size_t(cp_svd_me(m));
        return (me * n + me * me + me) * 8 + 512 + me * 4 + 1024;
    }
    bool take(cp_ctx *ctx, int m, int n) {
        const size_t me = size_t(cp_svd_me(m));
        Wk = cp_arena_take_t<double>(ctx, me * n);
        R = cp_arena_take_t<double>(ctx, me * me);
        sig = cp_arena_take_t<double>(ctx, me);
        rotated = cp_arena_take_t<int>(ctx, 128);   // [0] per-sweep counter, [4..5] the norm floor, [16..] the one-launch control block
        order = cp_arena_take_t<int>(ctx, me);
        return Wk && R && sig && rotated && order;
    }
};

// sign_ns.hip: projector onto the r leading eigenvectors by the Newton-Schulz sign iteration; the threshold is carried from
// call to call as a fraction of the trace
struct SignTracker {
    double sigma_rel = 0;   // threshold / trace(A) of the last call that found one (seeded by the caller)
    int steps_plan = 18;    // steps to run before the first look at the residuals
    long steps = 0;         // totals: Newton-Schulz steps, thresholds tried
    int trials = 0;
};
size_t cp_sign_workspace_doubles(int np_);
int cp_sign_projector(cp_ctx *ctx, const double *A, int np_, int r, SignTracker &tk, double *work, double *P, bool *found);

int cp_svd_rows_impl(cp_ctx *ctx, const double *M, int ldm, int m, int n, int r, double *sigma, double *Vt, int ldv,
                     double *SH, int ldsh, SvdScratch &sc, int *sweeps_out);
int cp_svd_rows_core(cp_ctx *ctx, const double *M, int ldm, int m, int n, int r, double *sigma, double *Vt, int ldv,
                     double *SH, int ldsh, SvdScratch &sc, int *sweeps_out, bool preinit, double rel_floor, double tol_in);
