// cp_prune_layer: the device side of one dictionary() call (lib/decompose.py:386-634) behind a
// single C entry -- LASSO operands, the whole alpha search, the mask, the least-squares refit and
// the copies back -- so that a host thread spends one foreign call (no interpreter work, no lock
// held) per layer.  Composition of the public entry points; no arithmetic of its own.
#include "cp_common.h"

#include <algorithm>
#include <chrono>
#include <memory>
#include <vector>

namespace {

int layer_ws_reserve(cp_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->layer_ws_bytes) return CP_OK;
    if (ctx->layer_ws) {
        CP_HIP(ctx, cp_stream_wait(ctx));
        CP_HIP(ctx, hipFree(ctx->layer_ws));
        ctx->layer_ws = nullptr;
        ctx->layer_ws_bytes = 0;
    }
    bytes = cp_align_up(bytes, 1 << 16);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->layer_ws), bytes);
    if (e != hipSuccess)
        return cp_set_error(ctx, CP_ERR_NOMEM, "layer workspace hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    ctx->layer_ws_bytes = bytes;
    return CP_OK;
}

// The overlapped normal equations (cp_refit_precompute_enqueue) belong to ONE cp_prune_layer(s) call: only the refit that call
// issues may consume them, and whatever way the call ends nothing of them stays pending -- a later refit on pooled buffers
// with the same addresses must never pick up a stale Gram.
struct PrecomputeScope {
    cp_ctx *ctx;
    explicit PrecomputeScope(cp_ctx *c) : ctx(c) {
        cp_precompute_void(ctx);      // left over from a call that failed midway
        ctx->pre.armed = true;
    }
    ~PrecomputeScope() {
        ctx->pre.armed = false;
        cp_precompute_void(ctx);      // no-op after a refit that consumed it
    }
};

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// host-side wall time of the last cp_prune_layer on this context (not part of the public ABI):
// {LASSO operands, alpha search, refit: host work only (time blocked on the stream excluded); final copies + wait}
extern "C" int cp_debug_host_times(cp_ctx *ctx, double *out4) {
    if (!ctx || !out4) return CP_ERR_ARG;
    for (int i = 0; i < 4; ++i) out4[i] = ctx->host_ms[i];
    return CP_OK;
}

// The (b, W) of the last cp_prune_layer / cp_prune_layers job on this context, where the last refit kernel left them:
// the context's pinned host block.  Valid until the next call on the context.
extern "C" int cp_result_host(cp_ctx *ctx, const double **b, const double **W, int *n, int *p) {
    if (!ctx || !b || !W || !n || !p) return CP_ERR_ARG;
    if (!ctx->pinned || ctx->result_n <= 0) return cp_set_error(ctx, CP_ERR_ARG, "cp_result_host: no result on this context");
    const double *b_host = reinterpret_cast<const double *>(ctx->pinned + 64);
    *b = b_host;
    *W = b_host + ctx->result_n;
    *n = ctx->result_n;
    *p = ctx->result_p;
    return CP_OK;
}

namespace {

// Operands of a dictionary() call that still sit in the caller's (pageable) host arrays: cp_prune_layer_h2d.
struct HostOperands {
    const void *X = nullptr;      // [N, c, kk], x_dtype
    const double *Y = nullptr;    // [N, n]
};

size_t elt_size(int dtype) { return dtype == CP_F32 ? 4 : 8; }

// The sampled rows of X and Y, gathered on the host into page-locked memory, go to the device first (a few MB): the LASSO
// operands and the alpha search -- milliseconds of one workgroup -- need nothing else.  -> compact device copies.
int stage_sampled_rows(cp_ctx *ctx, const HostOperands &h, int x_dtype, int c, int kk, int n, const int64_t *samples, int S,
                       void *Xs_dev, double *Ys_dev) {
    const size_t xrow = size_t(c) * kk * elt_size(x_dtype), yrow = size_t(n) * 8;
    const size_t need = size_t(S) * (xrow + yrow);
    if (need > ctx->stage_bytes) {
        if (ctx->stage) {
            CP_HIP(ctx, cp_stream_wait(ctx));
            CP_HIP(ctx, hipHostFree(ctx->stage));
            ctx->stage = nullptr;
            ctx->stage_bytes = 0;
        }
        CP_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->stage), cp_align_up(need, 1 << 16), hipHostMallocDefault));
        ctx->stage_bytes = cp_align_up(need, 1 << 16);
    }
    char *xs = ctx->stage, *ys = ctx->stage + size_t(S) * xrow;
    const char *X = static_cast<const char *>(h.X);
    const char *Y = reinterpret_cast<const char *>(h.Y);
    for (int s = 0; s < S; ++s) {
        memcpy(xs + size_t(s) * xrow, X + size_t(samples[s]) * xrow, xrow);
        memcpy(ys + size_t(s) * yrow, Y + size_t(samples[s]) * yrow, yrow);
    }
    CP_HIP(ctx, hipMemcpyAsync(Xs_dev, xs, size_t(S) * xrow, hipMemcpyHostToDevice, ctx->stream));
    CP_HIP(ctx, hipMemcpyAsync(Ys_dev, ys, size_t(S) * yrow, hipMemcpyHostToDevice, ctx->stream));
    return CP_OK;
}

// One dictionary() call.  host.X / host.Y null: X and Y are resident (cp_prune_layer).  Otherwise X / Y are the device
// buffers the call FILLS from the host arrays, behind the alpha search:
//     own stream    sampled rows (page-locked, a few MB) -> LASSO operands -> alpha search (async)
//     side stream   X, Y from the caller's pageable arrays (the host thread stages them while the search runs), then the
//                   overlapped normal equations of the refit (cp_refit_precompute_enqueue) when the flags ask for them
//     own stream    wait for the search, mask, refit (ordered after the uploads)
int prune_layer_impl(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2, int w_dtype, int n,
                     const double *Y, const HostOperands &host, const int64_t *samples, int S, double alpha_right0, double rank,
                     double lbound, double rbound, const uint32_t *seeds, int max_fits, int max_iter, double tol, int flags,
                     double ridge, uint8_t *mask_out, double *W_out, double *b_out, cp_prune_result *res) {
    if (!ctx || !X || !W2 || !Y || !mask_out || !res || (W_out == nullptr) != (b_out == nullptr))
        return ctx ? cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer: null argument") : CP_ERR_ARG;
    if (c <= 0 || n <= 0 || kk <= 0 || N <= 0 || max_fits < 0 || max_fits > CP_MAX_FITS)
        return cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer: bad size (c=%d n=%d kk=%d N=%lld max_fits=%d)", c, n, kk,
                            (long long)N, max_fits);
    const bool streamed = host.X != nullptr;
    if (streamed && !host.Y) return cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer_h2d: Y_host missing");
    CP_HIP(ctx, hipSetDevice(ctx->device));
    PrecomputeScope pre_scope(ctx);
    memset(res, 0, sizeof(*res));
    ctx->result_n = 0;
    const size_t cc = size_t(c);
    const size_t n_q = cp_align_up(cc * cc, 32), n_v = cp_align_up(cc, 32), n_w = cp_align_up(size_t(n) * cc * kk, 32),
                 n_b = cp_align_up(size_t(n), 32);
    const size_t n_xs = streamed ? cp_align_up(size_t(std::max(S, 1)) * cc * kk, 32) : 0,
                 n_ys = streamed ? cp_align_up(size_t(std::max(S, 1)) * n, 32) : 0;
    CP_TRY(layer_ws_reserve(ctx, (n_q + 3 * n_v + n_w + n_b + n_xs + n_ys) * sizeof(double)));
    double *Q = reinterpret_cast<double *>(ctx->layer_ws);
    double *q = Q + n_q, *stats = q + n_v, *w = stats + n_v, *Wd = w + n_v, *bd = Wd + n_w;
    double *Xs_dev = bd + n_b, *Ys_dev = Xs_dev + n_xs;
    const size_t x_bytes = size_t(N) * cc * kk * elt_size(x_dtype), y_bytes = size_t(N) * n * 8;
    hipStream_t side = streamed ? cp_side_stream(ctx) : nullptr;
    // The copies out of the caller's pageable arrays must not outlive the call: the Python side drops its references to
    // those arrays when the call raises.  Whatever way the function returns after upload_rest(), the side stream's copies
    // are complete (a successful return has waited on ctx->stream, which is ordered behind ev_upload; an error return
    // synchronises with the event here).
    struct UploadJoin {
        cp_ctx *ctx;
        bool enqueued = false;
        ~UploadJoin() {
            if (enqueued && ctx->ev_upload) (void)hipEventSynchronize(ctx->ev_upload);
        }
    } upload_join{ctx};
    res->uploaded = 0;
    auto upload_rest = [&]() -> int {   // X, Y on the side stream; the host thread stages the pageable arrays meanwhile
        if (ctx->pre.ready && (X == ctx->pre.X || Y == ctx->pre.Y)) cp_precompute_void(ctx);   // new contents
        // a copy may be in flight out of the caller's pageable array when a later step fails: that return joins the side
        // stream itself (UploadJoin only knows about copies an event was recorded behind), and `uploaded` stays 0 -- the
        // device buffers are partially overwritten and the Python side re-arms the upload
        struct SideJoin {
            hipStream_t side;
            bool armed = true;
            ~SideJoin() {
                if (armed) (void)hipStreamSynchronize(side);
            }
        } side_join{side};
        CP_HIP(ctx, hipMemcpyAsync(const_cast<void *>(X), host.X, x_bytes, hipMemcpyHostToDevice, side));
        CP_HIP(ctx, hipMemcpyAsync(const_cast<double *>(Y), host.Y, y_bytes, hipMemcpyHostToDevice, side));
        if (!ctx->ev_upload) CP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
        CP_HIP(ctx, hipEventRecord(ctx->ev_upload, side));
        side_join.armed = false;
        upload_join.enqueued = true;
        res->uploaded = 1;            // from here on X_dev / Y_dev hold (or are about to hold) the caller's arrays
        return CP_OK;
    };

    std::vector<double> w_host(cc);
    if (rank >= double(c)) {  // decompose.py:487-488: nothing to select
        for (size_t i = 0; i < cc; ++i) mask_out[i] = 1;
        res->fits_used = 0;
        res->alpha = 0.0;
        if (streamed) {
            CP_TRY(upload_rest());
            CP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload, 0));
        }
    } else {
        if (!samples || S <= 0 || !seeds || max_fits == 0)
            return cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer: samples / seeds missing");
        const double t0 = now_ms(), w0 = ctx->wait_ms;
        const bool precompute = (flags & CP_REFIT_PRECOMPUTE) && ridge == 0.0;
        int fits_used = 0;
        double alpha = 0.0;
        int rc;
        if (!streamed) {
            CP_TRY(cp_lasso_gram(ctx, X, x_dtype, N, c, kk, W2, w_dtype, n, Y, samples, S, Q, q, stats));
            if (precompute) CP_TRY(cp_refit_precompute_enqueue(ctx, X, x_dtype, N, c, kk, Y, n, (flags & CP_REFIT_PREFACTOR) ? rank : 0.0));
            ctx->host_ms[0] = now_ms() - t0;
            rc = cp_lasso_alpha_search(ctx, Q, c, q, stats, c, double(S) * double(n), alpha_right0, rank, lbound, rbound, seeds,
                                       max_fits, max_iter, tol, flags, w, &fits_used, &alpha, res->fit_log, res->fit_alpha);
        } else {
            for (int s = 0; s < S; ++s)
                if (samples[s] < 0 || samples[s] >= N) return cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer_h2d: sample out of range");
            CP_TRY(stage_sampled_rows(ctx, host, x_dtype, c, kk, n, samples, S, Xs_dev, Ys_dev));
            std::vector<int64_t> iota(static_cast<size_t>(S), 0);
            for (int s = 0; s < S; ++s) iota[size_t(s)] = s;
            // the same rows in the same order as cp_lasso_gram(X, samples) reads them: identical Q, q, stats
            CP_TRY(cp_lasso_gram(ctx, Xs_dev, x_dtype, S, c, kk, W2, w_dtype, n, Ys_dev, iota.data(), S, Q, q, stats));
            cp_search_job sj;
            sj.Q = Q; sj.ldq = c; sj.q = q; sj.stats = stats; sj.c = c; sj.M = double(S) * double(n);
            sj.alpha_right0 = alpha_right0; sj.rank = rank; sj.lbound = lbound; sj.rbound = rbound; sj.seeds = seeds;
            sj.max_fits = max_fits; sj.max_iter = max_iter; sj.tol = tol; sj.flags = flags; sj.w = w;
            cp_ctx *one[1] = {ctx};
            if (precompute) {   // what the side stream's work has to wait for on this stream ends here, BEFORE the search
                if (!ctx->ev_fork) {
                    CP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                    CP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
                }
                CP_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            }
            CP_TRY(cp_alpha_search_enqueue_batch(one, 1, &sj));   // asynchronous: the uploads below run under it
            CP_TRY(upload_rest());
            if (precompute)   // on the side stream, i.e. behind the uploads
                CP_TRY(cp_refit_precompute_enqueue(ctx, X, x_dtype, N, c, kk, Y, n, (flags & CP_REFIT_PREFACTOR) ? rank : 0.0, true));
            ctx->host_ms[0] = now_ms() - t0;
            bool timed_out = false;
            CP_HIP(ctx, cp_stream_wait(ctx));
            rc = cp_alpha_search_collect(ctx, c, max_fits, &fits_used, &alpha, res->fit_log, res->fit_alpha, &timed_out);
            if (timed_out && cp_cd_kernel_form(c, flags) == CP_CD_FORM_MULTI) {   // the one-workgroup team, bit-identical
                ++ctx->cd_fallbacks;
                CP_TRY(cp_alpha_search_enqueue_batch(one, 1, &sj, false));
                CP_HIP(ctx, cp_stream_wait(ctx));
                rc = cp_alpha_search_collect(ctx, c, max_fits, &fits_used, &alpha, res->fit_log, res->fit_alpha, &timed_out);
            }
            CP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload, 0));   // whatever follows on this stream reads X, Y
        }
        const double t1 = t0 + ctx->host_ms[0];
        if (rc == CP_ERR_NUMERIC) {  // did not settle within max_fits: the caller replays fit by fit
            res->fits_used = -1;
            if (streamed) CP_HIP(ctx, cp_stream_wait(ctx));   // X, Y complete before the caller touches them
            return CP_OK;
        }
        CP_TRY(rc);
        res->fits_used = fits_used;
        res->alpha = alpha;
        memcpy(w_host.data(), ctx->pinned_w, cc * sizeof(double));  // written by the search kernel itself
        for (size_t i = 0; i < cc; ++i) mask_out[i] = w_host[i] != 0.0 ? 1 : 0;  // decompose.py:463
        ctx->host_ms[1] = now_ms() - t1 - (ctx->wait_ms - w0);  // host work only (enqueue + copies)
    }
    int nnz = 0;
    for (size_t i = 0; i < cc; ++i) nnz += mask_out[i];
    res->nnz = nnz;
    cp_refit_info info;
    const double t2 = now_ms(), w2 = ctx->wait_ms;
    CP_TRY(cp_lstsq_refit_impl(ctx, X, x_dtype, N, c, kk, mask_out, Y, n, ridge, Wd, bd, &info, true));
    const double t3 = now_ms();
    ctx->host_ms[2] = t3 - t2 - (ctx->wait_ms - w2);  // host work only
    res->p = info.p;
    res->refit_rank = info.rank;
    res->fallback = info.fallback;
    // the last kernel of the refit wrote b and W into the pinned block itself
    const double *b_host = reinterpret_cast<const double *>(ctx->pinned + 64);
    ctx->result_n = n;
    ctx->result_p = info.p;
    if (W_out) {  // else: borrowed through cp_result_host
        memcpy(b_out, b_host, size_t(n) * sizeof(double));
        memcpy(W_out, b_host + n, size_t(n) * size_t(info.p) * sizeof(double));
    }
    ctx->host_ms[3] = now_ms() - t3;  // copies back + last wait
    return CP_OK;
}

}  // namespace

extern "C" int cp_prune_layer(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2,
                              int w_dtype, int n, const double *Y, const int64_t *samples, int S,
                              double alpha_right0, double rank, double lbound, double rbound,
                              const uint32_t *seeds, int max_fits, int max_iter, double tol, int flags,
                              double ridge, uint8_t *mask_out, double *W_out, double *b_out,
                              cp_prune_result *res) {
    return prune_layer_impl(ctx, X, x_dtype, N, c, kk, W2, w_dtype, n, Y, HostOperands{}, samples, S, alpha_right0, rank, lbound,
                            rbound, seeds, max_fits, max_iter, tol, flags, ridge, mask_out, W_out, b_out, res);
}

extern "C" int cp_prune_layer_h2d(cp_ctx *ctx, void *X_dev, const void *X_host, int x_dtype, int64_t N, int c, int kk,
                                  const void *W2, int w_dtype, int n, double *Y_dev, const double *Y_host,
                                  const int64_t *samples, int S, double alpha_right0, double rank, double lbound,
                                  double rbound, const uint32_t *seeds, int max_fits, int max_iter, double tol, int flags,
                                  double ridge, uint8_t *mask_out, double *W_out, double *b_out, cp_prune_result *res) {
    if (!X_host || !Y_host) return ctx ? cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer_h2d: host operands missing") : CP_ERR_ARG;
    HostOperands h;
    h.X = X_host;
    h.Y = Y_host;
    return prune_layer_impl(ctx, X_dev, x_dtype, N, c, kk, W2, w_dtype, n, Y_dev, h, samples, S, alpha_right0, rank, lbound, rbound,
                            seeds, max_fits, max_iter, tol, flags, ridge, mask_out, W_out, b_out, res);
}


// cp_prune_layers: the same work for several independent layers of the same width that share ONE stream.  A stream
// runs one kernel at a time and a layer is mostly its single-workgroup alpha search, so the searches of the batch
// are the workgroups of one launch; LASSO operands and refits are enqueued layer after layer around it.  Two host
// waits per batch (after the searches, after the refits).  Contexts: one per job (own arena / pinned block), all
// bound to the same stream (cp_ctx_set_stream), distinct from each other.
extern "C" int cp_prune_layers(int n_jobs, cp_ctx *const *ctxs, const cp_prune_job *jobs, cp_prune_result *results) {
    if (!ctxs || !jobs || !results || n_jobs <= 0 || !ctxs[0]) return CP_ERR_ARG;
    cp_ctx *ctx0 = ctxs[0];
    if (n_jobs > CP_MAX_JOBS) return cp_set_error(ctx0, CP_ERR_ARG, "cp_prune_layers: at most %d jobs per call (got %d)", CP_MAX_JOBS, n_jobs);
    for (int l = 0; l < n_jobs; ++l) {
        const cp_prune_job &j = jobs[l];
        if (!ctxs[l] || !j.X || !j.W2 || !j.Y || !j.mask_out || (j.W_out == nullptr) != (j.b_out == nullptr))
            return cp_set_error(ctx0, CP_ERR_ARG, "cp_prune_layers: null argument in job %d", l);
        if (ctxs[l]->stream != ctx0->stream || ctxs[l]->device != ctx0->device)
            return cp_set_error(ctx0, CP_ERR_ARG, "cp_prune_layers: the contexts of a batch must share one stream "
                                                  "(cp_ctx_set_stream) and device (job %d)", l);
        for (int m = 0; m < l; ++m)
            if (ctxs[m] == ctxs[l]) return cp_set_error(ctx0, CP_ERR_ARG, "cp_prune_layers: context used twice");
        if (j.c != jobs[0].c || j.c <= 0 || j.n <= 0 || j.kk <= 0 || j.N <= 0 || j.max_fits < 0 || j.max_fits > CP_MAX_FITS)
            return cp_set_error(ctx0, CP_ERR_ARG, "cp_prune_layers: job %d: bad size or channel count differs from job 0", l);
    }
    CP_HIP(ctx0, hipSetDevice(ctx0->device));
    std::vector<std::unique_ptr<PrecomputeScope>> pre_scopes;
    for (int l = 0; l < n_jobs; ++l) pre_scopes.emplace_back(new PrecomputeScope(ctxs[l]));
    for (int l = 0; l < n_jobs; ++l) ctxs[l]->refit_pending = false;  // nothing left over from a call that failed midway
    const int c = jobs[0].c;
    const size_t cc = size_t(c);
    struct Ws {
        double *Q, *q, *stats, *w, *Wd, *bd;
    } ws[CP_MAX_JOBS];
    cp_search_job sj[CP_MAX_JOBS];
    cp_ctx *sctx[CP_MAX_JOBS];
    int smap[CP_MAX_JOBS], n_search = 0;
    for (int l = 0; l < n_jobs; ++l) {
        const cp_prune_job &j = jobs[l];
        cp_ctx *ctx = ctxs[l];
        memset(&results[l], 0, sizeof(cp_prune_result));
        const size_t n_q = cp_align_up(cc * cc, 32), n_v = cp_align_up(cc, 32),
                     n_w = cp_align_up(size_t(j.n) * cc * j.kk, 32), n_b = cp_align_up(size_t(j.n), 32);
        CP_TRY(layer_ws_reserve(ctx, (n_q + 3 * n_v + n_w + n_b) * sizeof(double)));
        Ws &w = ws[l];
        w.Q = reinterpret_cast<double *>(ctx->layer_ws);
        w.q = w.Q + n_q;
        w.stats = w.q + n_v;
        w.w = w.stats + n_v;
        w.Wd = w.w + n_v;
        w.bd = w.Wd + n_w;
        if (j.rank >= double(c)) {  // decompose.py:487-488: nothing to select
            for (size_t i = 0; i < cc; ++i) j.mask_out[i] = 1;
            continue;
        }
        if (!j.samples || j.S <= 0 || !j.seeds || j.max_fits == 0)
            return cp_set_error(ctx0, CP_ERR_ARG, "cp_prune_layers: samples / seeds missing in job %d", l);
        CP_TRY(cp_lasso_gram(ctx, j.X, j.x_dtype, j.N, c, j.kk, j.W2, j.w_dtype, j.n, j.Y, j.samples, j.S, w.Q, w.q, w.stats));
        if ((j.flags & CP_REFIT_PRECOMPUTE) && j.ridge == 0.0) {
            const int rc = cp_refit_precompute_enqueue(ctx, j.X, j.x_dtype, j.N, c, j.kk, j.Y, j.n);
            if (rc != CP_OK) {
                if (ctx != ctx0) cp_set_error(ctx0, rc, "cp_prune_layers: job %d: %s", l, ctx->err);
                return rc;
            }
        }
        cp_search_job &s = sj[n_search];
        s.Q = w.Q; s.ldq = c; s.q = w.q; s.stats = w.stats; s.c = c; s.M = double(j.S) * double(j.n);
        s.alpha_right0 = j.alpha_right0; s.rank = j.rank; s.lbound = j.lbound; s.rbound = j.rbound; s.seeds = j.seeds;
        s.max_fits = j.max_fits; s.max_iter = j.max_iter; s.tol = j.tol; s.flags = j.flags; s.w = w.w;
        sctx[n_search] = ctx;
        smap[n_search++] = l;
    }
    if (n_search > 0) {
        CP_TRY(cp_alpha_search_enqueue_batch(sctx, n_search, sj));
        CP_HIP(ctx0, cp_stream_wait(sctx[0]));   // one wait for every search of the batch
        // a search whose multi-CU team reported a hand-off time-out runs again on the one-workgroup team (bit-identical)
        int n_retry = 0, rmap[CP_MAX_JOBS];
        cp_search_job rj[CP_MAX_JOBS];
        cp_ctx *rctx[CP_MAX_JOBS];
        for (int pass = 0; pass < 2; ++pass) {
            const int count = pass == 0 ? n_search : n_retry;
            int next_retry = 0;
            for (int k = 0; k < count; ++k) {
                const int ks = pass == 0 ? k : rmap[k];
                const int l = smap[ks];
                const cp_prune_job &j = jobs[l];
                int fits_used = 0;
                double alpha = 0.0;
                bool timed_out = false;
                const int rc = cp_alpha_search_collect(sctx[ks], c, j.max_fits, &fits_used, &alpha, results[l].fit_log,
                                                       results[l].fit_alpha, &timed_out);
                if (timed_out && pass == 0 && cp_cd_kernel_form(c, j.flags) == CP_CD_FORM_MULTI) {
                    rj[next_retry] = sj[ks];
                    rctx[next_retry] = sctx[ks];
                    rmap[next_retry++] = ks;
                    ++sctx[ks]->cd_fallbacks;
                    continue;
                }
                if (rc == CP_ERR_NUMERIC) {  // did not settle within max_fits: the caller replays this layer fit by fit
                    results[l].fits_used = -1;
                    continue;
                }
                results[l].fits_used = fits_used;
                results[l].alpha = alpha;
                const double *wh = sctx[ks]->pinned_w;
                for (size_t i = 0; i < cc; ++i) j.mask_out[i] = wh[i] != 0.0 ? 1 : 0;  // decompose.py:463
            }
            n_retry = next_retry;
            if (pass == 0 && n_retry > 0) {
                CP_TRY(cp_alpha_search_enqueue_batch(rctx, n_retry, rj, false));
                CP_HIP(ctx0, cp_stream_wait(rctx[0]));
            } else {
                break;
            }
        }
    }
    // refits: every layer's front (means, centring, Gram, X^T Y) is enqueued, then the batch factors and substitutes
    cp_refit_info info[CP_MAX_JOBS];
    for (int l = 0; l < n_jobs; ++l) {
        if (results[l].fits_used < 0) continue;
        const cp_prune_job &j = jobs[l];
        cp_ctx *ctx = ctxs[l];
        int nnz = 0;
        for (size_t i = 0; i < cc; ++i) nnz += j.mask_out[i];
        results[l].nnz = nnz;
        ctx->defer_refit_wait = true;
        ctx->refit_pending = false;
        const int rc = cp_lstsq_refit_impl(ctx, j.X, j.x_dtype, j.N, c, j.kk, j.mask_out, j.Y, j.n, j.ridge, ws[l].Wd, ws[l].bd,
                                           &info[l], true);
        ctx->defer_refit_wait = false;
        if (rc != CP_OK) {
            if (ctx != ctx0) cp_set_error(ctx0, rc, "cp_prune_layers: job %d: %s", l, ctx->err);
            return rc;
        }
    }
    CP_TRY(cp_refit_batch_factor_solve(ctxs, n_jobs));  // every pending factorisation in one launch, every substitution in another
    CP_HIP(ctx0, cp_stream_wait(ctx0));
    for (int l = 0; l < n_jobs; ++l) {
        if (results[l].fits_used < 0) continue;
        const cp_prune_job &j = jobs[l];
        cp_ctx *ctx = ctxs[l];
        if (ctx->refit_pending) {
            ctx->refit_pending = false;
            if (*reinterpret_cast<const int *>(ctx->pinned) != 0) {  // a pivot failed: the rank-deficient branch, synchronously
                const int rc = cp_lstsq_refit_impl(ctx, j.X, j.x_dtype, j.N, c, j.kk, j.mask_out, j.Y, j.n, j.ridge, ws[l].Wd,
                                                   ws[l].bd, &info[l], true);
                if (rc != CP_OK) {
                    if (ctx != ctx0) cp_set_error(ctx0, rc, "cp_prune_layers: job %d: %s", l, ctx->err);
                    return rc;
                }
            }
        }
        results[l].p = info[l].p;
        results[l].refit_rank = info[l].rank;
        results[l].fallback = info[l].fallback;
        const double *b_host = reinterpret_cast<const double *>(ctx->pinned + 64);
        ctx->result_n = j.n;
        ctx->result_p = info[l].p;
        if (j.W_out) {
            memcpy(j.b_out, b_host, size_t(j.n) * sizeof(double));
            memcpy(j.W_out, b_host + j.n, size_t(j.n) * size_t(info[l].p) * sizeof(double));
        }
    }
    return CP_OK;
}
