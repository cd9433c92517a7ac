// cp_prune_layer: the device side of one dictionary() call (lib/decompose.py:386-634) behind a
// single C entry -- LASSO operands, the whole alpha search, the mask, the least-squares refit and
// the copies back -- so that a host thread spends one foreign call (no interpreter work, no lock
// held) per layer.  Composition of the public entry points; no arithmetic of its own.
#include "cp_common.h"

#include <chrono>

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

extern "C" int cp_prune_layer(cp_ctx *ctx, const void *X, int x_dtype, int64_t N, int c, int kk, const void *W2,
                              int w_dtype, int n, const double *Y, const int64_t *samples, int S,
                              double alpha_right0, double rank, double lbound, double rbound,
                              const uint32_t *seeds, int max_fits, int max_iter, double tol, int flags,
                              double ridge, uint8_t *mask_out, double *W_out, double *b_out,
                              cp_prune_result *res) {
    if (!ctx || !X || !W2 || !Y || !mask_out || !W_out || !b_out || !res)
        return ctx ? cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer: null argument") : CP_ERR_ARG;
    if (c <= 0 || n <= 0 || kk <= 0 || N <= 0 || max_fits < 0 || max_fits > CP_MAX_FITS)
        return cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer: bad size (c=%d n=%d kk=%d N=%lld max_fits=%d)", c, n, kk,
                            (long long)N, max_fits);
    CP_HIP(ctx, hipSetDevice(ctx->device));
    memset(res, 0, sizeof(*res));
    const size_t cc = size_t(c);
    const size_t n_q = cp_align_up(cc * cc, 32), n_v = cp_align_up(cc, 32), n_w = cp_align_up(size_t(n) * cc * kk, 32),
                 n_b = cp_align_up(size_t(n), 32);
    CP_TRY(layer_ws_reserve(ctx, (n_q + 3 * n_v + n_w + n_b) * sizeof(double)));
    double *Q = reinterpret_cast<double *>(ctx->layer_ws);
    double *q = Q + n_q, *stats = q + n_v, *w = stats + n_v, *Wd = w + n_v, *bd = Wd + n_w;

    std::vector<double> w_host(cc);
    if (rank >= double(c)) {  // decompose.py:487-488: nothing to select
        for (size_t i = 0; i < cc; ++i) mask_out[i] = 1;
        res->fits_used = 0;
        res->alpha = 0.0;
    } else {
        if (!samples || S <= 0 || !seeds || max_fits == 0)
            return cp_set_error(ctx, CP_ERR_ARG, "cp_prune_layer: samples / seeds missing");
        const double t0 = now_ms(), w0 = ctx->wait_ms;
        CP_TRY(cp_lasso_gram(ctx, X, x_dtype, N, c, kk, W2, w_dtype, n, Y, samples, S, Q, q, stats));
        const double t1 = now_ms();
        ctx->host_ms[0] = t1 - t0;
        int fits_used = 0;
        double alpha = 0.0;
        const int rc = cp_lasso_alpha_search(ctx, Q, c, q, stats, c, double(S) * double(n), alpha_right0, rank, lbound,
                                             rbound, seeds, max_fits, max_iter, tol, flags, w, &fits_used, &alpha,
                                             res->fit_log, res->fit_alpha);
        if (rc == CP_ERR_NUMERIC) {  // did not settle within max_fits: the caller replays fit by fit
            res->fits_used = -1;
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
    memcpy(b_out, b_host, size_t(n) * sizeof(double));
    memcpy(W_out, b_host + n, size_t(n) * size_t(info.p) * sizeof(double));
    ctx->host_ms[3] = now_ms() - t3;  // copies back + last wait
    return CP_OK;
}
