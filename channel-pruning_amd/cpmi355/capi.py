"""ctypes binding of libcpmi355.so (C ABI declared in include/cpmi355.h).

This is the only place the shared library is loaded.  There is NO CPU fallback: if the
library is missing or no gfx950 device is visible, loading / context creation raises, and
every product entry point (lib/decompose.py, lib/net.py) propagates that error.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcpmi355.so")

CP_ERR_NOMEM = -3          # include/cpmi355.h
CP_F32, CP_F64 = 0, 1
CP_CD_RECIPROCAL = 1
CP_CD_DELTA = 2
CP_REFIT_PRECOMPUTE = 4
CP_REFIT_PREFACTOR = 8
CP_MAX_STAGES = 32

_c_int, _c_i64, _c_dbl, _c_u32 = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_uint32
_vp = ctypes.c_void_p


class CdResult(ctypes.Structure):
    _fields_ = [("gap", _c_dbl), ("tol_scaled", _c_dbl), ("n_iter", ctypes.c_int32), ("nnz", ctypes.c_int32),
                ("edge_margin", _c_dbl), ("gap_margin", _c_dbl)]


class RefitInfo(ctypes.Structure):
    _fields_ = [("p", ctypes.c_int32), ("rank", ctypes.c_int32), ("fallback", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


CP_MAX_FITS = 64


class PruneResult(ctypes.Structure):
    _fields_ = [("fits_used", ctypes.c_int32), ("nnz", ctypes.c_int32), ("p", ctypes.c_int32),
                ("refit_rank", ctypes.c_int32), ("fallback", ctypes.c_int32), ("uploaded", ctypes.c_int32),
                ("alpha", ctypes.c_double), ("fit_log", CdResult * CP_MAX_FITS),
                ("fit_alpha", ctypes.c_double * CP_MAX_FITS)]


class PruneJob(ctypes.Structure):
    """cp_prune_job: the argument list of cp_prune_layer as one record (cp_prune_layers takes an array of them)."""
    _fields_ = [("X", _vp), ("x_dtype", ctypes.c_int32), ("c", ctypes.c_int32), ("N", ctypes.c_int64),
                ("kk", ctypes.c_int32), ("w_dtype", ctypes.c_int32), ("W2", _vp), ("n", ctypes.c_int32),
                ("S", ctypes.c_int32), ("Y", _vp), ("samples", _vp), ("alpha_right0", _c_dbl), ("rank", _c_dbl),
                ("lbound", _c_dbl), ("rbound", _c_dbl), ("seeds", _vp), ("max_fits", ctypes.c_int32),
                ("max_iter", ctypes.c_int32), ("tol", _c_dbl), ("flags", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("ridge", _c_dbl), ("mask_out", _vp), ("W_out", _vp), ("b_out", _vp)]


class CpError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__("libcpmi355: %s failed: %s (%d)%s" % (what, _strerror(code), code,
                                                              (" -- " + detail) if detail else ""))


# name -> (restype, argtypes); must list every symbol include/cpmi355.h declares
SIGNATURES = {
    "cp_version": (_c_int, []),
    "cp_strerror": (ctypes.c_char_p, [_c_int]),
    "cp_last_error": (ctypes.c_char_p, [_vp]),
    "cp_device_count": (_c_int, [ctypes.POINTER(_c_int)]),
    "cp_ctx_create": (_c_int, [_c_int, ctypes.POINTER(_vp)]),
    "cp_ctx_create_priority": (_c_int, [_c_int, _c_int, ctypes.POINTER(_vp)]),
    "cp_ctx_destroy": (_c_int, [_vp]),
    "cp_ctx_create_sibling": (_c_int, [_vp, ctypes.POINTER(_vp)]),
    "cp_ctx_set_stream": (_c_int, [_vp, _vp]),
    "cp_sync": (_c_int, [_vp]),
    "cp_malloc": (_c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "cp_free": (_c_int, [_vp, _vp]),
    "cp_memcpy_h2d": (_c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "cp_memcpy_d2h": (_c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "cp_memset": (_c_int, [_vp, _vp, _c_int, ctypes.c_size_t]),
    "cp_patch_gather": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int,
                                 _c_int, _c_int, _vp, _c_i64]),
    "cp_patch_gather_batches": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int,
                                         _c_int, _c_int, _vp]),
    "cp_assemble_y": (_c_int, [_vp, _vp, _vp, _vp, _c_i64, _c_int, _vp]),
    "cp_lasso_gram": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_int,
                               _vp, _vp, _vp]),
    "cp_cd_kernel_form": (_c_int, [_c_int, _c_int]),
    "cp_enet_cd_gram": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _c_int, _c_dbl, _c_dbl, _c_u32, _c_int, _c_dbl,
                                 _c_int, _vp, ctypes.POINTER(CdResult)]),
    "cp_lasso_alpha_search": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _c_int, _c_dbl, _c_dbl, _c_dbl, _c_dbl, _c_dbl,
                                       _vp, _c_int, _c_int, _c_dbl, _c_int, _vp, ctypes.POINTER(_c_int),
                                       ctypes.POINTER(_c_dbl), _vp, _vp]),
    "cp_lstsq_refit": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _vp, _c_int, _c_dbl, _vp, _vp,
                                ctypes.POINTER(RefitInfo)]),
    "cp_refit_shard_layout": (_c_int, [_c_int, _c_int, _c_int, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64)]),
    "cp_refit_shard_sums": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _vp, _c_int, _vp]),
    "cp_refit_shard_gram": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _vp, _c_int, _c_i64, _vp, _vp]),
    "cp_refit_shard_solve": (_c_int, [_vp, _c_int, _c_int, _c_int, _c_i64, _c_dbl, _vp, _vp, _vp, _vp,
                                      ctypes.POINTER(RefitInfo)]),
    "cp_nonlinear_fc": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _c_int, _vp, _vp,
                                 ctypes.POINTER(RefitInfo)]),
    "cp_svd_rows": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, ctypes.POINTER(_c_int)]),
    "cp_svd_rows_lowrank": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, ctypes.POINTER(_c_int)]),
    "cp_vh_project": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _c_int, _vp, _c_int, _vp]),
    "cp_matmul_tn": (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _c_int, _vp]),
    "cp_itq_iterate": (_c_int, [_vp, _vp, _vp, _c_i64, _c_int, _c_int, _vp, _vp, _c_int, _c_dbl, _vp, _vp, _vp]),
    "cp_prune_layer": (_c_int, [_vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_int,
                                _c_dbl, _c_dbl, _c_dbl, _c_dbl, _vp, _c_int, _c_int, _c_dbl, _c_int, _c_dbl,
                                _vp, _vp, _vp, ctypes.POINTER(PruneResult)]),
    "cp_prune_layer_h2d": (_c_int, [_vp, _vp, _vp, _c_int, _c_i64, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _vp, _c_int,
                                    _c_dbl, _c_dbl, _c_dbl, _c_dbl, _vp, _c_int, _c_int, _c_dbl, _c_int, _c_dbl,
                                    _vp, _vp, _vp, ctypes.POINTER(PruneResult)]),
    "cp_prune_layers": (_c_int, [_c_int, ctypes.POINTER(_vp), _vp, _vp]),
    "cp_result_host": (_c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_c_int),
                                ctypes.POINTER(_c_int)]),
    "cp_probe_mfma_f64": (_c_int, [_vp, ctypes.POINTER(_c_dbl)]),
    "cp_probe_mfma_f64_clock": (_c_int, [_vp, ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_dbl), ctypes.POINTER(_c_dbl)]),
    "cp_probe_hbm_copy": (_c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_c_dbl)]),
    "cp_debug_gemm_units": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int]),
    "cp_last_stage_times": (_c_int, [_vp, ctypes.POINTER(_c_int), ctypes.POINTER(ctypes.c_float)]),
    "cp_stage_name": (ctypes.c_char_p, [_vp, _c_int]),
    "cp_enable_stage_timing": (_c_int, [_vp, _c_int]),
    "cp_stage_epoch": (_c_int, [_vp]),
    "cp_last_stage_spans": (_c_int, [_vp, _vp, ctypes.POINTER(_c_int), ctypes.POINTER(ctypes.c_float),
                                     ctypes.POINTER(ctypes.c_float)]),
}

_lib = None


def load():
    """dlopen libcpmi355.so (never falls back to anything else)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise ImportError("libcpmi355.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C channel-pruning_amd/csrc` (expected at %s)" % LIB_PATH)
        # Independent layers run on their own HIP streams.  With the runtime's default of 4 hardware
        # queues several streams share one, and a multi-millisecond single-wave CD kernel then holds up
        # every kernel queued behind it; has to be in the environment before the HIP runtime starts.
        # 16, not more: the runtime never gives a hardware queue back, and a process that has once had more than
        # ~16 of them alive runs EVERYTHING afterwards slower -- the vgg16_5x job 24.7 ms in a fresh process,
        # 26.4 / 27.0 ms after a leg with 24 streams alive when the limit is 24, 27.8 / 28.0 ms when it is 32,
        # 24.7 ms when it is 16 (tools/probes/leg_residue.py).  The jobs themselves do not
        # care: vgg16 (12 streams) 24.2-24.6 ms at 12 / 16 / 24 / 32, resnet50 (24 streams) 32.7 against 32.1 ms.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        # torch ships its own libamdhip64 / libhsa-runtime64.  Whichever HIP runtime enters the process first serves
        # both; torch cannot initialise on the system one ("No HIP GPUs are available"), this library runs equally
        # fast on either (measured).  So a process that also uses torch.cuda (cpmi355.shard's row-sharded path,
        # bench.py --gpus > 1) has to import torch BEFORE the first Context; CP_PRELOAD_TORCH=1 does it here.
        if os.environ.get("CP_PRELOAD_TORCH", "") == "1":
            import torch  # noqa: F401
        lib = ctypes.CDLL(os.environ.get("CP_LIB_PATH") or LIB_PATH)   # CP_LIB_PATH: a variant build (kernel experiments)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _strerror(code):
    try:
        return load().cp_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "error"


def _ptr(x):
    """device pointer of a DevBuf / torch tensor / int; host pointer of a numpy array."""
    if x is None:
        return None
    if isinstance(x, DevBuf):
        return x.ptr
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class DevBuf:
    """A device allocation owned by a Context (cp_malloc / cp_free).  free() hands the block to the context's pool: the
    drop-in uploads the same operand sizes call after call, and hipMalloc / hipFree are device-wide synchronisations
    (1.5-2 ms per dictionary() call at c = 512, with occasional 60 ms outliers).  Everything that touched the block ran on
    the context's stream or was waited for by the call that used it, so the next owner's work is ordered behind it."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr = ctx._pool_take(self.nbytes)
        if self.ptr is None:
            p = _vp()
            rc = ctx.lib.cp_malloc(ctx.h, self.nbytes, ctypes.byref(p))
            if rc != 0:                     # out of memory: give the pooled blocks back and try once more
                ctx._pool_drain()
                ctx._check(ctx.lib.cp_malloc(ctx.h, self.nbytes, ctypes.byref(p)), "cp_malloc")
            self.ptr = p.value

    def free(self):
        if self.ptr and self.ctx.h:
            self.ctx._pool_give(self.ptr, self.nbytes)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One device + one stream (cp_ctx).  Create it inside the process that uses it."""

    def __init__(self, device=0, priority=None):
        """priority: HIP priority of the context's stream (< 0 = higher); None = the environment's CP_CTX_PRIORITY (default 0)"""
        self.lib = load()
        h = _vp()
        if priority is None:
            rc = self.lib.cp_ctx_create(int(device), ctypes.byref(h))
        else:
            rc = self.lib.cp_ctx_create_priority(int(device), int(priority), ctypes.byref(h))
        if rc != 0:
            raise CpError(rc, "cp_ctx_create(device=%d)" % device,
                          "a gfx950 (MI355X) device is required; there is no CPU fallback")
        self.h = h.value
        self.device = int(device)
        self.pid = os.getpid()

    def sibling(self):
        """A context with its own workspace on THIS context's stream (for cp_prune_layers batches); close it first."""
        other = Context.__new__(Context)
        other.lib = self.lib
        h = _vp()
        self._check(self.lib.cp_ctx_create_sibling(self.h, ctypes.byref(h)), "cp_ctx_create_sibling")
        other.h, other.device, other.pid = h.value, self.device, self.pid
        other._parent = self       # the stream belongs to `self`: keep it alive as long as the sibling
        return other

    def close(self):
        if getattr(self, "h", None) and self.pid == os.getpid():
            self._pool_drain()
            self.lib.cp_ctx_destroy(self.h)
        self.h = None

    # -- pool of freed device blocks (exact-size reuse; see DevBuf) ---------------------------
    POOL_LIMIT = int(os.environ.get("CP_POOL_BYTES", str(4 << 30)))

    def _pool_take(self, nbytes):
        blocks = self.__dict__.setdefault("_pool", {}).get(nbytes)
        if blocks:
            self._pool_bytes -= nbytes
            return blocks.pop()
        return None

    def _pool_give(self, ptr, nbytes):
        pool = self.__dict__.setdefault("_pool", {})
        if self.__dict__.setdefault("_pool_bytes", 0) + nbytes > self.POOL_LIMIT:
            self.lib.cp_free(self.h, ptr)
            return
        pool.setdefault(nbytes, []).append(ptr)
        self._pool_bytes += nbytes

    def _pool_drain(self):
        for blocks in self.__dict__.get("_pool", {}).values():
            for ptr in blocks:
                self.lib.cp_free(self.h, ptr)
        self._pool, self._pool_bytes = {}, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            err = CpError(rc, what, self.lib.cp_last_error(self.h).decode() if self.h else "")
            if rc == CP_ERR_NOMEM:      # an allocation inside the library failed: give the pooled blocks back, so that a
                self._pool_drain()      # retry by the caller does not fail next to gigabytes of idle blocks
            raise err

    # -- memory ---------------------------------------------------------------------
    def empty(self, nbytes):
        return DevBuf(self, nbytes)

    def to_device(self, arr, buf=None):
        arr = np.ascontiguousarray(arr)
        if buf is None:
            buf = DevBuf(self, max(arr.nbytes, 8))
        assert buf.nbytes >= arr.nbytes
        self._check(self.lib.cp_memcpy_h2d(self.h, buf.ptr, arr.ctypes.data, arr.nbytes), "cp_memcpy_h2d")
        return buf

    def to_host(self, buf, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._check(self.lib.cp_memcpy_d2h(self.h, out.ctypes.data, _ptr(buf), out.nbytes), "cp_memcpy_d2h")
        return out

    def zeros(self, nbytes):
        b = DevBuf(self, nbytes)
        self._check(self.lib.cp_memset(self.h, b.ptr, 0, nbytes), "cp_memset")
        return b

    def sync(self):
        self._check(self.lib.cp_sync(self.h), "cp_sync")

    def set_stream(self, stream_ptr):
        self._check(self.lib.cp_ctx_set_stream(self.h, stream_ptr), "cp_ctx_set_stream")

    # -- hot path -------------------------------------------------------------------
    def patch_gather(self, fmap, B, C, H, W, xs, ys, k, pad, stride, relu, X_out, row0):
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        ys = np.ascontiguousarray(ys, dtype=np.int32)
        self._check(self.lib.cp_patch_gather(self.h, _ptr(fmap), B, C, H, W, xs.ctypes.data, ys.ctypes.data,
                                             xs.shape[0], k, pad, stride, int(bool(relu)), _ptr(X_out),
                                             int(row0)), "cp_patch_gather")

    def patch_gather_batches(self, fmap, nb, B, C, H, W, xs, ys, P, k, pad, stride, relu, X_out):
        """all nb batches in one launch: fmap [nb,B,C,H,W] f32 on the device, xs / ys int32 [nb * P] batch-major"""
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        ys = np.ascontiguousarray(ys, dtype=np.int32)
        assert xs.shape[0] == nb * P and ys.shape[0] == nb * P
        self._check(self.lib.cp_patch_gather_batches(self.h, _ptr(fmap), int(nb), B, C, H, W, xs.ctypes.data, ys.ctypes.data,
                                                     int(P), k, pad, stride, int(bool(relu)), _ptr(X_out)),
                    "cp_patch_gather_batches")

    def assemble_y(self, feats, bias, resY, N, n, Y):
        self._check(self.lib.cp_assemble_y(self.h, _ptr(feats), _ptr(bias), _ptr(resY), int(N), int(n), _ptr(Y)),
                    "cp_assemble_y")

    def lasso_gram(self, X, x_dtype, N, c, kk, W2, w_dtype, n, Y, samples, Q, q, stats):
        samples = np.ascontiguousarray(samples, dtype=np.int64)
        self._check(self.lib.cp_lasso_gram(self.h, _ptr(X), x_dtype, int(N), int(c), int(kk), _ptr(W2), w_dtype,
                                           int(n), _ptr(Y), samples.ctypes.data, samples.shape[0], _ptr(Q),
                                           _ptr(q), _ptr(stats)), "cp_lasso_gram")

    def cd_kernel_form(self, c, flags=0):
        """0 one wavefront, 1 two waves, 2 team, 3 multi-CU team (include/cpmi355.h: CP_CD_FORM_*)."""
        return int(self.lib.cp_cd_kernel_form(int(c), int(flags)))

    def enet_cd_gram(self, Q, ldq, q, stats, c, l1_reg, l2_reg, seed, w, max_iter=1000, tol=1e-4, flags=0):
        res = CdResult()
        self._check(self.lib.cp_enet_cd_gram(self.h, _ptr(Q), int(ldq), _ptr(q), _ptr(stats), int(c),
                                             float(l1_reg), float(l2_reg), int(seed), int(max_iter), float(tol),
                                             int(flags), _ptr(w), ctypes.byref(res)), "cp_enet_cd_gram")
        return res

    def lasso_alpha_search(self, Q, ldq, q, stats, c, M, alpha_right0, rank, lbound, rbound, seeds, w,
                           max_iter=1000, tol=1e-4, flags=0):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        mf = seeds.shape[0]
        fits_used = _c_int()
        alpha_out = _c_dbl()
        log = (CdResult * mf)()
        alphas = np.zeros(mf, dtype=np.float64)
        self._check(self.lib.cp_lasso_alpha_search(self.h, _ptr(Q), int(ldq), _ptr(q), _ptr(stats), int(c),
                                                   float(M), float(alpha_right0), float(rank), float(lbound),
                                                   float(rbound), seeds.ctypes.data, mf, int(max_iter), float(tol),
                                                   int(flags), _ptr(w), ctypes.byref(fits_used),
                                                   ctypes.byref(alpha_out), ctypes.cast(log, _vp),
                                                   alphas.ctypes.data), "cp_lasso_alpha_search")
        nf = fits_used.value
        fits = [(float(alphas[i]), int(log[i].nnz), int(log[i].n_iter), float(log[i].edge_margin), float(log[i].gap_margin))
                for i in range(max(nf, 0))]
        return nf, alpha_out.value, fits

    def lstsq_refit(self, X, x_dtype, N, c, kk, mask, Y, n, ridge, W_out, b_out):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        info = RefitInfo()
        self._check(self.lib.cp_lstsq_refit(self.h, _ptr(X), x_dtype, int(N), int(c), int(kk), mask.ctypes.data,
                                            _ptr(Y), int(n), float(ridge), _ptr(W_out), _ptr(b_out),
                                            ctypes.byref(info)), "cp_lstsq_refit")
        return info

    # -- sample-sharded refit: three calls, the caller all-reduces `sums` and `gram` in between ---------
    def refit_shard_layout(self, kept, kk, n):
        a, b = _c_i64(), _c_i64()
        self._check(self.lib.cp_refit_shard_layout(int(kept), int(kk), int(n), ctypes.byref(a), ctypes.byref(b)),
                    "cp_refit_shard_layout")
        return a.value, b.value

    def refit_shard_sums(self, X, x_dtype, N_local, c, kk, mask, Y, n, sums):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.lib.cp_refit_shard_sums(self.h, _ptr(X), x_dtype, int(N_local), int(c), int(kk),
                                                 mask.ctypes.data, _ptr(Y), int(n), _ptr(sums)), "cp_refit_shard_sums")

    def refit_shard_gram(self, X, x_dtype, N_local, c, kk, mask, Y, n, N_total, sums, gram):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._check(self.lib.cp_refit_shard_gram(self.h, _ptr(X), x_dtype, int(N_local), int(c), int(kk),
                                                 mask.ctypes.data, _ptr(Y), int(n), int(N_total), _ptr(sums),
                                                 _ptr(gram)), "cp_refit_shard_gram")

    def refit_shard_solve(self, kept, kk, n, N_total, ridge, sums, gram, W_out, b_out):
        info = RefitInfo()
        self._check(self.lib.cp_refit_shard_solve(self.h, int(kept), int(kk), int(n), int(N_total), float(ridge),
                                                  _ptr(sums), _ptr(gram), _ptr(W_out), _ptr(b_out),
                                                  ctypes.byref(info)), "cp_refit_shard_solve")
        return info

    def nonlinear_fc(self, X, x_dtype, N, c, kk, mask, Y, n, W_out, b_out, iters=(30, 20), lambdas=(0.1, 1.0)):
        """ReLU-aware alternating reconstruction (cp_nonlinear_fc); defaults = the reference's schedule."""
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        it = np.ascontiguousarray(iters, dtype=np.int32)
        lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        info = RefitInfo()
        self._check(self.lib.cp_nonlinear_fc(self.h, _ptr(X), x_dtype, int(N), int(c), int(kk), mask.ctypes.data, _ptr(Y),
                                             int(n), it.ctypes.data, lam.ctypes.data, int(it.shape[0]), _ptr(W_out),
                                             _ptr(b_out), ctypes.byref(info)), "cp_nonlinear_fc")
        return info

    # -- VH_decompose pieces -------------------------------------------------------------
    def svd_rows(self, M, r, lowrank=False):
        """Leading r singular triplets of the host matrix M[m, n] (m <= n) -> (sigma[r], Vt[r, m], SH[r, n])
        with SH = diag(sigma) H; computed on the device (cp_svd_rows; lowrank: M has rank <= r, cp_svd_rows_lowrank)."""
        M = np.ascontiguousarray(M, dtype=np.float64)
        m, n = M.shape
        Md = self.to_device(M)
        sd, Vd, Hd = self.empty(r * 8), self.empty(r * m * 8), self.empty(r * n * 8)
        sweeps = _c_int()
        try:
            fn = self.lib.cp_svd_rows_lowrank if lowrank else self.lib.cp_svd_rows
            self._check(fn(self.h, Md.ptr, m, n, int(r), sd.ptr, Vd.ptr, Hd.ptr, ctypes.byref(sweeps)), "cp_svd_rows")
            self.last_svd_sweeps = sweeps.value
            return (self.to_host(sd, (r,), np.float64), self.to_host(Vd, (r, m), np.float64),
                    self.to_host(Hd, (r, n), np.float64))
        finally:
            for bfr in (Md, sd, Vd, Hd):
                bfr.free()

    def matmul_tn(self, A, B):
        """A[k, m]^T B[k, n] on the device (host arrays in, host array out)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        B = np.ascontiguousarray(B, dtype=np.float64)
        k, m = A.shape
        n = B.shape[1]
        Ad, Bd, Cd = self.to_device(A), self.to_device(B), self.empty(m * n * 8)
        try:
            self._check(self.lib.cp_matmul_tn(self.h, Ad.ptr, Bd.ptr, m, n, k, Cd.ptr), "cp_matmul_tn")
            return self.to_host(Cd, (m, n), np.float64)
        finally:
            for bfr in (Ad, Bd, Cd):
                bfr.free()

    def itq_iterate(self, feature, gt_feature, rank, iters=(30, 20), lambdas=(0.1, 1.0), pinv_cond=1e-6):
        """The alternations of ITQ_decompose on the device -> (T[n, n], Y_mean[n], U_mean[n]) as host arrays."""
        F = np.ascontiguousarray(feature, dtype=np.float64)
        Gt = np.ascontiguousarray(gt_feature, dtype=np.float64)
        N, n = F.shape
        it = np.ascontiguousarray(iters, dtype=np.int32)
        lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        Fd, Gd = self.to_device(F), self.to_device(Gt)
        Td, yd, ud = self.empty(n * n * 8), self.empty(n * 8), self.empty(n * 8)
        try:
            self._check(self.lib.cp_itq_iterate(self.h, Fd.ptr, Gd.ptr, N, n, int(rank), it.ctypes.data, lam.ctypes.data,
                                                int(it.shape[0]), float(pinv_cond), Td.ptr, yd.ptr, ud.ptr),
                        "cp_itq_iterate")
            return (self.to_host(Td, (n, n), np.float64), self.to_host(yd, (n,), np.float64),
                    self.to_host(ud, (n,), np.float64))
        finally:
            for bfr in (Fd, Gd, Td, yd, ud):
                bfr.free()

    def result_host(self):
        """(W [n, p], b [n]) of the last prune_layer on this context as VIEWS of the context's page-locked result
        block (cp_result_host): no copy, DMA-able; valid until the next call on the context / its close()."""
        pb, pw, n, p = _vp(), _vp(), _c_int(), _c_int()
        self._check(self.lib.cp_result_host(self.h, ctypes.byref(pb), ctypes.byref(pw), ctypes.byref(n),
                                            ctypes.byref(p)), "cp_result_host")
        b = np.frombuffer((ctypes.c_double * n.value).from_address(pb.value), dtype=np.float64)
        W = np.frombuffer((ctypes.c_double * (n.value * p.value)).from_address(pw.value), dtype=np.float64)
        return W.reshape(n.value, p.value), b

    def prune_layer(self, X, x_dtype, N, c, kk, W2, w_dtype, n, Y, samples, alpha_right0, rank, lbound, rbound,
                    seeds, ridge, flags=0, max_iter=1000, tol=1e-4, borrow=False, X_host=None, Y_host=None):
        """One dictionary() worth of device work in a single foreign call (cp_prune_layer).
        -> (PruneResult, mask bool[c], W f64[n, p], b f64[n]); res.fits_used == -1: search did not settle.
        borrow: W, b are result_host() views instead of fresh arrays.
        X_host / Y_host (C-contiguous host arrays): X / Y are device buffers the call fills from them, the upload running
        behind the alpha search (cp_prune_layer_h2d)."""
        samples = np.ascontiguousarray(samples, dtype=np.int64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        mask = np.zeros(int(c), dtype=np.uint8)
        res = PruneResult()
        if X_host is not None:
            W = None if borrow else np.empty(int(n) * int(c) * int(kk), dtype=np.float64)
            b = None if borrow else np.empty(int(n), dtype=np.float64)
            try:
                self._check(self.lib.cp_prune_layer_h2d(self.h, _ptr(X), X_host.ctypes.data, x_dtype, int(N), int(c), int(kk),
                                                        _ptr(W2), w_dtype, int(n), _ptr(Y), Y_host.ctypes.data,
                                                        samples.ctypes.data, int(samples.shape[0]), float(alpha_right0),
                                                        float(rank), float(lbound), float(rbound), seeds.ctypes.data,
                                                        int(seeds.shape[0]), int(max_iter), float(tol), int(flags),
                                                        float(ridge), mask.ctypes.data, None if borrow else W.ctypes.data,
                                                        None if borrow else b.ctypes.data, ctypes.byref(res)),
                            "cp_prune_layer_h2d")
            except CpError as e:
                e.uploaded = bool(res.uploaded)     # did X_dev / Y_dev receive the host arrays before the error?
                raise
            if res.fits_used < 0:
                return res, None, None, None
            if borrow:
                W, b = self.result_host()
                return res, mask.astype(bool), W, b
            return res, mask.astype(bool), W[:int(n) * int(res.p)].reshape(int(n), int(res.p)), b
        if borrow:
            self._check(self.lib.cp_prune_layer(self.h, _ptr(X), x_dtype, int(N), int(c), int(kk), _ptr(W2), w_dtype,
                                                int(n), _ptr(Y), samples.ctypes.data, int(samples.shape[0]),
                                                float(alpha_right0), float(rank), float(lbound), float(rbound),
                                                seeds.ctypes.data, int(seeds.shape[0]), int(max_iter), float(tol),
                                                int(flags), float(ridge), mask.ctypes.data, None, None,
                                                ctypes.byref(res)), "cp_prune_layer")
            if res.fits_used < 0:
                return res, None, None, None
            W, b = self.result_host()
            return res, mask.astype(bool), W, b
        W = np.empty(int(n) * int(c) * int(kk), dtype=np.float64)
        b = np.empty(int(n), dtype=np.float64)
        self._check(self.lib.cp_prune_layer(self.h, _ptr(X), x_dtype, int(N), int(c), int(kk), _ptr(W2), w_dtype,
                                            int(n), _ptr(Y), samples.ctypes.data, int(samples.shape[0]),
                                            float(alpha_right0), float(rank), float(lbound), float(rbound),
                                            seeds.ctypes.data, int(seeds.shape[0]), int(max_iter), float(tol),
                                            int(flags), float(ridge), mask.ctypes.data, W.ctypes.data, b.ctypes.data,
                                            ctypes.byref(res)), "cp_prune_layer")
        if res.fits_used < 0:
            return res, None, None, None
        return res, mask.astype(bool), W[:int(n) * int(res.p)].reshape(int(n), int(res.p)), b

    @staticmethod
    def prune_layers(jobs):
        """cp_prune_layers.  jobs: list of dicts with the keyword arguments of prune_layer plus "ctx" (distinct
        contexts on one stream: a Context and its sibling()s).  -> list of (PruneResult, mask, W, b) like prune_layer."""
        B = len(jobs)
        arr = (PruneJob * B)()
        ctxs = (_vp * B)()
        keep = []
        for i, j in enumerate(jobs):
            samples = np.ascontiguousarray(j["samples"], dtype=np.int64)
            seeds = np.ascontiguousarray(j["seeds"], dtype=np.uint32)
            c, n, kk = int(j["c"]), int(j["n"]), int(j["kk"])
            mask = np.zeros(c, dtype=np.uint8)
            borrow = bool(j.get("borrow", False))
            W = None if borrow else np.empty(n * c * kk, dtype=np.float64)
            b = None if borrow else np.empty(n, dtype=np.float64)
            keep.append((samples, seeds, mask, W, b))
            a = arr[i]
            a.X, a.x_dtype, a.c, a.N, a.kk = _ptr(j["X"]), j["x_dtype"], c, int(j["N"]), kk
            a.w_dtype, a.W2, a.n, a.S, a.Y = j["w_dtype"], _ptr(j["W2"]), n, samples.shape[0], _ptr(j["Y"])
            a.samples, a.alpha_right0, a.rank = samples.ctypes.data, float(j["alpha_right0"]), float(j["rank"])
            a.lbound, a.rbound, a.seeds = float(j["lbound"]), float(j["rbound"]), seeds.ctypes.data
            a.max_fits, a.max_iter, a.tol = seeds.shape[0], int(j.get("max_iter", 1000)), float(j.get("tol", 1e-4))
            a.flags, a.ridge = int(j.get("flags", 0)), float(j.get("ridge", 0.0))
            a.mask_out = mask.ctypes.data
            a.W_out, a.b_out = (None, None) if borrow else (W.ctypes.data, b.ctypes.data)
            ctxs[i] = j["ctx"].h
        res = (PruneResult * B)()
        ctx0 = jobs[0]["ctx"]
        ctx0._check(ctx0.lib.cp_prune_layers(B, ctxs, ctypes.cast(arr, _vp), ctypes.cast(res, _vp)), "cp_prune_layers")
        out = []
        for i, (samples, seeds, mask, W, b) in enumerate(keep):
            r = res[i]
            if r.fits_used < 0:
                out.append((r, None, None, None))
            elif W is None:
                Wv, bv = jobs[i]["ctx"].result_host()
                out.append((r, mask.astype(bool), Wv, bv))
            else:
                n, p = int(jobs[i]["n"]), int(r.p)
                out.append((r, mask.astype(bool), W[:n * p].reshape(n, p), b))
        return out

    # -- measurement ------------------------------------------------------------------
    def probe_mfma_f64(self):
        v = _c_dbl()
        self._check(self.lib.cp_probe_mfma_f64(self.h, ctypes.byref(v)), "cp_probe_mfma_f64")
        return v.value

    def probe_mfma_f64_clock(self):
        """-> (TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64, shader clock GHz during the launch, cycles per MFMA and SIMD)"""
        t, g, c = _c_dbl(), _c_dbl(), _c_dbl()
        self._check(self.lib.cp_probe_mfma_f64_clock(self.h, ctypes.byref(t), ctypes.byref(g), ctypes.byref(c)),
                    "cp_probe_mfma_f64_clock")
        return t.value, g.value, c.value

    def probe_hbm_copy(self, nbytes=1 << 30):
        v = _c_dbl()
        self._check(self.lib.cp_probe_hbm_copy(self.h, int(nbytes), ctypes.byref(v)), "cp_probe_hbm_copy")
        return v.value

    def enable_stage_timing(self, on=True):
        """False/0 off, True/1 every stage, 2 only the stage of the roofline kernel ("refit_gram_gemm")."""
        self._check(self.lib.cp_enable_stage_timing(self.h, int(on)), "cp_enable_stage_timing")

    def last_stage_times(self):
        cnt = _c_int()
        ms = (ctypes.c_float * CP_MAX_STAGES)()
        self._check(self.lib.cp_last_stage_times(self.h, ctypes.byref(cnt), ms), "cp_last_stage_times")
        return [(self.lib.cp_stage_name(self.h, i).decode(), float(ms[i])) for i in range(cnt.value)]

    def stage_epoch(self):
        """the reference event of last_stage_spans(epoch_of=self): recorded on this context's stream now"""
        self._check(self.lib.cp_stage_epoch(self.h), "cp_stage_epoch")

    def last_stage_spans(self, epoch_of):
        """-> [(stage name, ms, begin in ms after epoch_of.stage_epoch())]: the brackets of several contexts on one clock"""
        cnt = _c_int()
        ms = (ctypes.c_float * CP_MAX_STAGES)()
        begin = (ctypes.c_float * CP_MAX_STAGES)()
        self._check(self.lib.cp_last_stage_spans(self.h, epoch_of.h, ctypes.byref(cnt), ms, begin), "cp_last_stage_spans")
        return [(self.lib.cp_stage_name(self.h, i).decode(), float(ms[i]), float(begin[i])) for i in range(cnt.value)]


def device_count():
    n = _c_int()
    load().cp_device_count(ctypes.byref(n))
    return n.value


_default_ctx = {}


def default_context(device=None):
    """Per-process (fork-safe) default context on `device` (default: CP_DEVICE or 0)."""
    if device is None:
        device = int(os.environ.get("CP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (os.getpid(), device)
    ctx = _default_ctx.get(key)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[key] = ctx
    return ctx
