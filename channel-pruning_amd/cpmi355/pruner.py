"""Device-resident driver of one channel-pruning problem (one producer/consumer conv pair).

Mirrors the control flow of the reference's ``dictionary()`` (lib/decompose.py:386-634)
with the arithmetic on the GPU.  Everything that touches NumPy's RNG stays on the host and
consumes the stream exactly as the reference does:
  draw 1        samples = rng.randint(0, N, min(400, N // 20))          decompose.py:425
  draw 2..F+1   one seed per Lasso.fit: rng.randint(0, 2147483647)      _cd_fast.pyx:164
The alpha search itself (bracket doubling + bisection, decompose.py:490-525) runs either
  mode="device": one kernel launch for the whole search; the host pre-draws MAX_FITS seeds,
                 then rewinds the RNG and re-draws exactly the number of seeds the search
                 consumed, so the RNG state after the call equals the reference's;
  mode="host":   one launch per fit, the host deciding the next alpha (debug / logging).
"""
import ctypes

import numpy as np

from . import capi

RAND_R_MAX = 2147483647  # sklearn/linear_model/_cd_fast.pyx:26
MAX_FITS = 64


def precompute_flag(latency_mode, rank=None, c=None):
    """CP_REFIT_PRECOMPUTE: the normal equations over ALL c channels are computed on the device's side stream WHILE the
    (single-workgroup) alpha search runs; the refit then gathers the kept rows / columns and goes straight to the
    factorisation.  It costs (c / kept)^2 times the flops of the masked Gram, which is free when ONE layer has the chip
    to itself (the drop-in dictionary(): measured 18.8 -> 16.5 ms for a c = 512 layer) and a loss when many layers
    already fill it (the vgg16 job: 34.9 -> 37.2 ms) -- so: on for single-layer calls, off for batches / resident sets.
    Only when at least 70 % of the channels are to be kept (full Gram <= 2x the masked one): at rank = c / 2 the 4x Gram
    running next to the search slows the search itself (it reads Q from L2 every step: 96 -> 120 ns per step, conv3_x block
    7.6 -> 10.3 ms).  CP_REFIT_PRECOMPUTE=0 / 1 in the environment forces it.
    latency_mode True additionally sets CP_REFIT_PREFACTOR (the library applies it from rank >= 0.8 c): the FULL Gram is
    factored during the search as well and the refit becomes a constrained solve with that factor -- 2.4 instead of 6.2 ms
    between the end of the search and the result at c = 512, while the search itself runs 30 % slower next to the busy
    chip (9.1 -> 12.0 ms): 16.4 -> 15.1 ms per c = 512 layer, 6.7 -> 6.3 ms at c = 256.  latency_mode "gram": the normal
    equations only -- what a resident set uses for its heaviest layers (with the factorisation too: job 32.7 -> 34.0 ms)."""
    import os
    force = os.environ.get("CP_REFIT_PRECOMPUTE", "")
    pre = capi.CP_REFIT_PRECOMPUTE | (capi.CP_REFIT_PREFACTOR if latency_mode is True else 0)
    if force in ("0", "1"):
        return pre if force == "1" else 0
    dense = rank is None or c is None or (rank < c and rank >= 0.7 * c)
    return pre if latency_mode and dense else 0


# ---- RNG bookkeeping on the host (it sits under the interpreter lock of every worker thread, so it has to be cheap) ----
_MT_STATE_BYTES = 624 * 4 + 4    # numpy's mt19937_state: uint32 key[624]; int pos


def _mt_state_address(rng):
    """Address of the MT19937 state behind a legacy generator (np.random module or RandomState), else None."""
    gen = np.random.mtrand._rand if rng is np.random else rng
    bg = getattr(gen, "_bit_generator", None)
    if bg is None or type(bg).__name__ != "MT19937":
        return None
    return bg.ctypes.state_address


def rng_mark(rng):
    """Remember where `rng` stands: a 2.5 KB copy of the Mersenne-Twister state (about 1 us; get_state() + set_state()
    cost about 150 us of interpreter time), or get_state() for any other generator."""
    addr = _mt_state_address(rng)
    if addr is None:
        return (None, rng.get_state())
    buf = ctypes.create_string_buffer(_MT_STATE_BYTES)
    ctypes.memmove(buf, addr, _MT_STATE_BYTES)
    return (addr, buf)


def rng_rewind(rng, mark):
    addr, saved = mark
    if addr is None:
        rng.set_state(saved)
    else:
        ctypes.memmove(addr, saved, _MT_STATE_BYTES)


def draw_seeds(rng, count):
    """`count` Lasso.fit seeds, the values and the generator state count calls of rng.randint(0, 2147483647) give
    (one vectorised call: 8 us instead of 250 for 64 seeds; equality is pinned by tests/test_host_logic.py)."""
    return np.asarray(rng.randint(0, RAND_R_MAX, size=int(count)), dtype=np.int64).astype(np.uint32)


def _np_dtype_code(arr):
    if arr.dtype == np.float32:
        return capi.CP_F32
    if arr.dtype == np.float64:
        return capi.CP_F64
    raise TypeError("X/W2 must be float32 or float64, got %s" % arr.dtype)


class LayerProblem:
    """X[N,c,k,k], W2[n,c,k,k], Y[N,n] resident in HBM plus the small per-layer outputs."""

    @classmethod
    def from_device(cls, ctx, Xd, x_dtype, N, c, k, W2, Yd, flags=0):
        """X[N,c,k,k] (x_dtype) and Y[N,n] float64 already resident (DevBuf / data_ptr owner);
        W2 is the small host array [n,c,k,k]."""
        self = cls.__new__(cls)
        self.ctx = ctx
        W2 = np.ascontiguousarray(W2)
        if W2.dtype not in (np.float32, np.float64):
            W2 = W2.astype(np.float64)
        self.N, self.c, self.k, self.kk = int(N), int(c), int(k), int(k) * int(k)
        self.n = int(W2.shape[0])
        if W2.shape[1] != self.c or int(np.prod(W2.shape[2:])) != self.kk:
            raise ValueError("inconsistent shapes W2%s for c=%d k=%d" % (W2.shape, c, k))
        self.x_dtype, self.w_dtype = x_dtype, _np_dtype_code(W2)
        self.h2d_bytes = W2.nbytes
        self.Xd, self.Yd = Xd, Yd
        self._borrowed = ("Xd", "Yd")
        self._pending = None
        self.W2d = ctx.to_device(W2)
        self._alloc_outputs(flags)
        return self

    def _alloc_outputs(self, flags):
        ctx, c, n = self.ctx, self.c, self.n
        self.Qd = ctx.empty(c * c * 8)
        self.qd = ctx.empty(c * 8)
        self.statsd = ctx.empty(4 * 8)
        self.wd = ctx.zeros(c * 8)
        self.Wout = ctx.empty(n * c * self.kk * 8)
        self.bout = ctx.empty(n * 8)
        self.flags = flags
        self.S = 0
        self.fits = []          # [(alpha, nnz, n_iter)] of the last search
        self.margins = []       # [(edge_margin, gap_margin)] per fit: cp_cd_result's tie sentinels (-1 = not tracked)
        self.refit_info = None

    def __init__(self, ctx, X, W2, Y, flags=0, defer_upload=False):
        """defer_upload: X and Y get their device buffers but stay on the host until the first prune_fused(), which streams
        them in behind the alpha search (cp_prune_layer_h2d); any other use uploads them first (ensure_resident)."""
        self.ctx = ctx
        self._borrowed = ()
        self._pending = None
        X = np.ascontiguousarray(X)
        W2 = np.ascontiguousarray(W2)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        if W2.dtype not in (np.float32, np.float64):
            W2 = W2.astype(np.float64)
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.N, self.c = int(X.shape[0]), int(X.shape[1])
        self.kk = int(np.prod(X.shape[2:])) if X.ndim > 2 else 1
        self.k = int(X.shape[2]) if X.ndim > 2 else 1
        self.n = int(W2.shape[0])
        if W2.shape[1] != self.c or int(np.prod(W2.shape[2:])) != self.kk or Y.shape != (self.N, self.n):
            raise ValueError("inconsistent shapes X%s W2%s Y%s" % (X.shape, W2.shape, Y.shape))
        self.x_dtype, self.w_dtype = _np_dtype_code(X), _np_dtype_code(W2)
        self.h2d_bytes = X.nbytes + W2.nbytes + Y.nbytes
        if defer_upload:
            self.Xd = ctx.empty(X.nbytes)
            self.Yd = ctx.empty(Y.nbytes)
            self._pending = (X, Y)          # keeps the host arrays alive until they are on the device
        else:
            self.Xd = ctx.to_device(X)
            self.Yd = ctx.to_device(Y)
        self.W2d = ctx.to_device(W2)
        self._alloc_outputs(flags)

    def ensure_resident(self):
        """X and Y on the device (a no-op unless the upload was deferred and nothing has streamed them in yet)"""
        if self._pending is not None:
            X, Y = self._pending
            self._pending = None
            self.ctx._check(self.ctx.lib.cp_memcpy_h2d(self.ctx.h, self.Xd.ptr, X.ctypes.data, X.nbytes), "cp_memcpy_h2d")
            self.ctx._check(self.ctx.lib.cp_memcpy_h2d(self.ctx.h, self.Yd.ptr, Y.ctypes.data, Y.nbytes), "cp_memcpy_h2d")

    # -- decompose.py:425-437 + Lasso.fit preprocessing ------------------------------
    def lasso_gram(self, samples):
        self.ensure_resident()
        self.S = int(len(samples))
        self.ctx.lasso_gram(self.Xd, self.x_dtype, self.N, self.c, self.kk, self.W2d, self.w_dtype, self.n,
                            self.Yd, samples, self.Qd, self.qd, self.statsd)

    @property
    def M(self):
        return float(self.S * self.n)

    def reset_w(self):
        self.ctx._check(self.ctx.lib.cp_memset(self.ctx.h, self.wd.ptr, 0, self.c * 8), "cp_memset")

    # -- decompose.py:453-466: one solve(alpha) -----------------------------------------
    def solve(self, alpha, seed):
        r = self.ctx.enet_cd_gram(self.Qd, self.c, self.qd, self.statsd, self.c, alpha * self.M, 0.0, seed,
                                  self.wd, flags=self.flags)
        self.fits.append((float(alpha), int(r.nnz), int(r.n_iter)))
        self.margins.append((float(r.edge_margin), float(r.gap_margin)))
        return int(r.nnz)

    def mask(self):
        w = self.ctx.to_host(self.wd, (self.c,), np.float64)
        return w != 0.

    @staticmethod
    def rank_bounds(rank, rank_tol):
        """Acceptance window of the alpha search (decompose.py:493-501)."""
        lbound = rank
        if rank_tol >= 1:
            rbound = rank + rank_tol
        else:
            rbound = rank + rank_tol * rank
            if rank_tol == .2:            # decompose.py:498-501
                lbound = rank + 0.1 * rank
                rbound = rank + 0.2 * rank
        return lbound, rbound

    # -- the whole call in one foreign call (cp_prune_layer) ------------------------------
    def prune_fused(self, rank, alpha_right0, rank_tol, rng, samples, ridge=0.0, latency_mode=True):
        """-> (idxs, W[n,p], b, alpha) or None when the device search did not settle within MAX_FITS
        (the RNG is then back where it was before the seeds were drawn)."""
        lbound, rbound = self.rank_bounds(rank, rank_tol)
        self.S = int(len(samples))
        mark = rng_mark(rng)
        seeds = draw_seeds(rng, MAX_FITS)
        pending, self._pending = self._pending, None
        try:
            res, idxs, W, b = self.ctx.prune_layer(self.Xd, self.x_dtype, self.N, self.c, self.kk, self.W2d, self.w_dtype,
                                                   self.n, self.Yd, samples, alpha_right0, rank, lbound, rbound, seeds,
                                                   ridge, flags=self.flags | precompute_flag(latency_mode, rank, self.c),
                                                   borrow=getattr(self, "borrow_results", False),
                                                   X_host=pending[0] if pending else None,
                                                   Y_host=pending[1] if pending else None)
        except BaseException as e:   # noqa
            # an error return that came before the streamed upload was enqueued (bad argument, allocation failure ...)
            # leaves Xd / Yd empty: the host arrays stay pending, so a later refit() / lasso_gram() uploads them instead
            # of reading uninitialised device memory (cp_prune_result.uploaded).  The same for anything that is not an
            # error return of the library (a ctypes.ArgumentError, KeyboardInterrupt ...): nothing says the arrays arrived
            if pending is not None and not getattr(e, "uploaded", False):
                self._pending = pending
            rng_rewind(rng, mark)
            raise
        rng_rewind(rng, mark)
        if res.fits_used < 0:
            return None
        draw_seeds(rng, res.fits_used)     # consume exactly what the reference would have
        self.fits = [(float(res.fit_alpha[i]), int(res.fit_log[i].nnz), int(res.fit_log[i].n_iter))
                     for i in range(res.fits_used)]
        self.margins = [(float(res.fit_log[i].edge_margin), float(res.fit_log[i].gap_margin)) for i in range(res.fits_used)]
        info = capi.RefitInfo()
        info.p, info.rank, info.fallback = res.p, res.refit_rank, res.fallback
        self.refit_info = info
        return idxs, W, b, float(res.alpha)

    # -- decompose.py:490-525 -----------------------------------------------------------
    def alpha_search(self, rank, alpha_right0, rank_tol, rng, mode="device"):
        lbound, rbound = self.rank_bounds(rank, rank_tol)
        self.fits = []
        self.margins = []
        self.reset_w()
        if mode == "device":
            mark = rng_mark(rng)
            seeds = draw_seeds(rng, MAX_FITS)
            try:
                nf, alpha, fits = self.ctx.lasso_alpha_search(self.Qd, self.c, self.qd, self.statsd, self.c, self.M,
                                                              alpha_right0, rank, lbound, rbound, seeds, self.wd,
                                                              flags=self.flags)
            except capi.CpError as e:
                if e.code != -5:
                    raise
                nf = -1
            rng_rewind(rng, mark)
            if nf > 0:
                draw_seeds(rng, nf)        # consume exactly what the reference would have
                self.fits = [f[:3] for f in fits]
                self.margins = [f[3:] for f in fits]
                return alpha
            self.reset_w()                 # did not settle within MAX_FITS: replay fit by fit
        left, right = 0, alpha_right0
        while True:
            tmp = self.solve(right, rng.randint(0, RAND_R_MAX))
            if tmp < rank:
                break
            right *= 2
        while True:
            alpha = (left + right) / 2
            tmp = self.solve(alpha, rng.randint(0, RAND_R_MAX))
            if tmp > rbound:
                left = alpha
            elif tmp < lbound:
                right = alpha
            else:
                break
        return alpha

    # -- decompose.py:622-623 -> fc_kernel ------------------------------------------------
    def refit(self, idxs, ridge=0.0):
        self.ensure_resident()
        idxs = np.asarray(idxs, dtype=bool)
        info = self.ctx.lstsq_refit(self.Xd, self.x_dtype, self.N, self.c, self.kk, idxs.astype(np.uint8), self.Yd,
                                    self.n, ridge, self.Wout, self.bout)
        self.refit_info = info
        p = int(info.p)
        W = self.ctx.to_host(self.Wout, (self.n, p), np.float64)
        b = self.ctx.to_host(self.bout, (self.n,), np.float64)
        return W, b

    # -- decompose.py:615-617 -> nonlinear_fc ------------------------------------------------
    def refit_nonlinear(self, idxs, iters=(30, 20), lambdas=(0.1, 1.0)):
        self.ensure_resident()
        idxs = np.asarray(idxs, dtype=bool)
        info = self.ctx.nonlinear_fc(self.Xd, self.x_dtype, self.N, self.c, self.kk, idxs.astype(np.uint8), self.Yd,
                                     self.n, self.Wout, self.bout, iters=iters, lambdas=lambdas)
        self.refit_info = info
        p = int(info.p)
        return (self.ctx.to_host(self.Wout, (self.n, p), np.float64), self.ctx.to_host(self.bout, (self.n,), np.float64))

    def free(self):
        for name in ("Xd", "W2d", "Yd", "Qd", "qd", "statsd", "wd", "Wout", "bout"):
            buf = getattr(self, name, None)
            if buf is not None and name not in self._borrowed and hasattr(buf, "free"):
                buf.free()


GLOBAL_RNG = np.random  # module-level legacy RandomState: randint / get_state / set_state

# A decision whose relative margin is below this is reported (tie_report).  A HEURISTIC, not a bound: the device solves the
# Gram-form recurrence, the reference scikit-learn's DATA form; the two differ by about eps (|q_i| + |H_i|) times the
# accumulation growth over the S n sampled rows, which is not relative to alpha -- where q_i and H_i cancel (|q_i|, |H_i| >>
# alpha) the reference can flip a decision at a margin far above a few ulp of alpha.  2^20 eps = 2.3e-10 of alpha leaves room
# for |q_i| + |H_i| up to ~1e4 alpha at 64 ulp of accumulated error; no reference golden comes that close (asserted by
# tests/test_gpu_parity.py::_check_against_golden).
# Only the dead-zone edge and the duality-gap stop are watched; the d_w_max / w_max < tol early-exit test is not.
TIE_MARGIN = 2.0 ** 20 * float(np.finfo(np.float64).eps)


def tie_report(prob):
    """The tie sentinels of the last search on `prob` (cp_cd_result.edge_margin / gap_margin per fit) -> dict
        edge_margin   smallest relative distance of a coefficient, at its last update of a fit, from the edge of its dead zone
        gap_margin    smallest relative distance of a duality gap from its stopping threshold
        suspect       True when either is below TIE_MARGIN (a heuristic, see there): the reference (scikit-learn's DATA form
                      of the recurrence, lib/decompose.py:449, 456) may have decided that coefficient / that stop the other
                      way, so the mask is not pinned by rounding-level agreement alone
        tracked       False when the kernel form does not report the sentinels (-1)."""
    edges = [m[0] for m in prob.margins if m[0] >= 0]
    gaps = [m[1] for m in prob.margins if m[1] >= 0]
    edge = min(edges) if edges else None
    gap = min(gaps) if gaps else None
    return dict(edge_margin=edge, gap_margin=gap, tracked=bool(edges or gaps),
                suspect=bool((edge is not None and edge <= TIE_MARGIN) or (gap is not None and gap <= TIE_MARGIN)))


def prune_layer(prob, rank, alpha_in, rank_tol=.1, rng=None, ridge=0.0, mode="device", alpha_arg=1e-4,
                refit="linear", W2_host=None, latency_mode=True, fixed_alpha=None):
    """dictionary() on a resident LayerProblem -> (idxs, newW2[n,nnz,k,k], newB2, alpha_out).

    refit: "linear" (fc_kernel), "nonlinear" (nonlinear_fc) or "none" (dcfgs.nofc: W2[:, idxs], zero bias).
    latency_mode: this layer has the GPU to itself (see precompute_flag); callers that keep many layers in flight pass False.
    fixed_alpha: dcfgs.autodet (decompose.py:395-416, 582-585): ONE fit at that alpha decides the kept channels, no search.
    mode "device": the whole call is ONE foreign call (cp_prune_layer; linear refit only); "steps": the same device
    search through the individual entry points (lasso_gram / alpha search / refit); "host": one
    launch per LASSO fit, the host deciding the next alpha."""
    rng = GLOBAL_RNG if rng is None else rng
    N, c, n, k = prob.N, prob.c, prob.n, prob.k
    samples = rng.randint(0, N, min(400, N // 20))               # decompose.py:425
    prob.samples = samples
    if fixed_alpha is not None:                                   # decompose.py:582-585: idxs, rank = solve(alpha)
        prob.lasso_gram(samples)
        prob.fits = []
        prob.margins = []
        prob.reset_w()
        prob.solve(fixed_alpha, rng.randint(0, RAND_R_MAX))
        idxs = prob.mask()
        alpha = fixed_alpha
    elif rank == c:                                               # decompose.py:487-488
        idxs = np.array([True] * rank)
        alpha = alpha_arg
        prob.fits = []
        prob.margins = []
    else:
        if mode == "device" and refit == "linear":
            fused = prob.prune_fused(rank, alpha_in, rank_tol, rng, samples, ridge, latency_mode=latency_mode)
            if fused is not None:
                idxs, W, b, alpha = fused
                return idxs, W.reshape((n, int(idxs.sum()), k, k)), b, alpha
            mode = "host"                                         # did not settle within MAX_FITS
        prob.lasso_gram(samples)
        alpha = prob.alpha_search(rank, alpha_in, rank_tol, rng, mode="device" if mode == "steps" else mode)
        idxs = prob.mask()
    nnz = int(idxs.sum())
    if refit == "none":                                           # dcfgs.nofc (decompose.py:618-620)
        return idxs, np.asarray(W2_host)[:, idxs, :, :], np.zeros(n), alpha
    if refit == "nonlinear":                                      # dcfgs.nonlinear_fc (decompose.py:615-617)
        W, b = prob.refit_nonlinear(idxs)
    else:
        W, b = prob.refit(idxs, ridge=ridge)
    return idxs, W.reshape((n, nnz, k, k)), b, alpha


def prune_layers_batched(probs, ranks, alpha_ins, rngs, rank_tol=.1, ridge=0.0, alpha_arg=1e-4):
    """dictionary() for several independent resident LayerProblems of the same channel count in ONE foreign call
    (cp_prune_layers): their alpha searches run side by side as the workgroups of one launch.  probs[i].ctx are
    distinct contexts on one stream (a Context and its sibling()s); rngs[i] is layer i's own RandomState-like
    generator (independent layers have independent draw sequences: each consumes exactly what the reference's
    dictionary() would -- the sample subset, then one seed per fit).  -> list of (idxs, newW2, newB2, alpha_out)."""
    B = len(probs)
    jobs, states, samples_l = [], [], []
    for prob, rank, alpha_in, rng in zip(probs, ranks, alpha_ins, rngs):
        samples = rng.randint(0, prob.N, min(400, prob.N // 20))                 # decompose.py:425
        prob.samples, prob.S = samples, int(len(samples))
        lbound, rbound = LayerProblem.rank_bounds(rank, rank_tol)
        states.append(rng_mark(rng))
        seeds = np.zeros(1, dtype=np.uint32) if rank == prob.c else draw_seeds(rng, MAX_FITS)
        samples_l.append(samples)
        jobs.append(dict(ctx=prob.ctx, X=prob.Xd, x_dtype=prob.x_dtype, N=prob.N, c=prob.c, kk=prob.kk, W2=prob.W2d,
                         w_dtype=prob.w_dtype, n=prob.n, Y=prob.Yd, samples=samples, alpha_right0=alpha_in, rank=rank,
                         lbound=lbound, rbound=rbound, seeds=seeds, ridge=ridge,
                         flags=prob.flags | precompute_flag(False), borrow=getattr(prob, "borrow_results", False)))
    raw = capi.Context.prune_layers(jobs)
    out = []
    for i, (prob, rank, alpha_in, rng) in enumerate(zip(probs, ranks, alpha_ins, rngs)):
        res, idxs, W, b = raw[i]
        rng_rewind(rng, states[i])
        n, k = prob.n, prob.k
        if res.fits_used < 0:                      # did not settle within MAX_FITS: replay this layer fit by fit
            prob.lasso_gram(samples_l[i])
            alpha = prob.alpha_search(rank, alpha_in, rank_tol, rng, mode="host")
            idxs = prob.mask()
            W, b = prob.refit(idxs, ridge=ridge)
            out.append((idxs, W.reshape((n, int(idxs.sum()), k, k)), b, alpha))
            continue
        draw_seeds(rng, res.fits_used)             # consume exactly what the reference would have
        prob.fits = [(float(res.fit_alpha[j]), int(res.fit_log[j].nnz), int(res.fit_log[j].n_iter))
                     for j in range(res.fits_used)]
        prob.margins = [(float(res.fit_log[j].edge_margin), float(res.fit_log[j].gap_margin)) for j in range(res.fits_used)]
        info = capi.RefitInfo()
        info.p, info.rank, info.fallback = res.p, res.refit_rank, res.fallback
        prob.refit_info = info
        alpha = float(res.alpha) if rank != prob.c else alpha_arg
        out.append((idxs, W.reshape((n, int(idxs.sum()), k, k)), b, alpha))
    return out
