"""Drop-in for the pruning entry points of the reference's lib/decompose.py, computed on
MI355X through libcpmi355.so (no scikit-learn / SciPy call on this path, no CPU fallback).

    dictionary(X, W2, Y, alpha=1e-4, rank=None, DEBUG=0, B2=None, rank_tol=.1, verbose=0)
        -> (idxs bool[c], newW2 float64[n, nnz, k, k], newB2 float64[n])     decompose.py:386-634
    fc_kernel(X, Y, copy_X=True, W=None, B=None, ret_reg=False, fit_intercept=True)
        -> (coef_[n, p], intercept_[n]) or an estimator-like object           decompose.py:636-669
    rel_error, relu                                                           decompose.py:22-32

Same argument meaning, same side effects as the reference: consumes numpy's GLOBAL RNG
(one draw for the sample subset + one per LASSO fit, decompose.py:425 / _cd_fast.pyx:164),
reads ``dcfgs.dic.rank_tol`` / ``dcfgs.fc_ridge`` / ``dcfgs.autodet`` and the module global
``cfgs.alpha``, writes ``cfgs.alpha`` back (decompose.py:626-627); inputs are not modified.
Branches of the reference that depend on modules absent from its own tree (lightning, keras,
theanols, GDsolver; decompose.py:12-20, 641-660) raise NotImplementedError here.
"""
import numpy as np

import warnings

from cpmi355 import LayerProblem, default_context, prune_layer
from cpmi355 import capi as _capi
from cpmi355.pruner import TIE_MARGIN, tie_report

from . import cfgs
from .cfgs import c as dcfgs

# filled by every dictionary() call: per-fit (alpha, nnz, n_iter), the sample subset, refit path
last_call_info = {}


def relu(x):
    return np.maximum(x, 0.)


def rel_error(A, B):
    return np.mean((A - B) ** 2) ** .5 / np.mean(A ** 2) ** .5


def _flags():
    return (_capi.CP_CD_RECIPROCAL if dcfgs.cd_reciprocal else 0) | (_capi.CP_CD_DELTA if dcfgs.cd_delta else 0)


def _refit_mode():
    """Validate the dcfgs flags of dictionary() (decompose.py:393-416, 605-623) and pick the refit branch."""
    if dcfgs.dic.alter or dcfgs.ls != 'linear' or dcfgs.solver != cfgs.solvers.sk or dcfgs.dic.debug:
        raise NotImplementedError("only the sklearn/linear configuration of the reference is accelerated "
                                  "(dic.alter=0, ls='linear', solver='sklearn')")
    if dcfgs.nonlinear_fc and dcfgs.fc_ridge:
        raise NotImplementedError("nonlinear_fc with fc_ridge > 0")
    return "nonlinear" if dcfgs.nonlinear_fc else ("none" if dcfgs.nofc else "linear")   # decompose.py:615-623


def prune_resident(prob, rank, W2_host, alpha=1e-4):
    """The body of dictionary() on operands that are already resident (a cpmi355.LayerProblem): used by dictionary()
    below and by Net.dictionary_kernel (lib/net.py), so both honour the same dcfgs flags (rank_tol, fc_ridge,
    nonlinear_fc, nofc, cd_mode), consume numpy's global RNG identically and carry cfgs.alpha (decompose.py:626-627)."""
    refit = _refit_mode()
    fixed = None
    if dcfgs.autodet:                                    # decompose.py:395-397, 414-415: no rank, one fit at a derived alpha
        fixed = cfgs.alpha / prob.c ** dcfgs.dic.layeralpha
    idxs, newW2, newB2, alpha_out = prune_layer(prob, rank, cfgs.alpha, rank_tol=dcfgs.dic.rank_tol, rng=np.random,
                                                ridge=float(dcfgs.fc_ridge), mode=dcfgs.cd_mode,
                                                alpha_arg=alpha, refit=refit, W2_host=W2_host, fixed_alpha=fixed)
    last_call_info.clear()
    ri = prob.refit_info
    ties = tie_report(prob)
    last_call_info.update(fits=list(prob.fits), samples=prob.samples,
                          fallback=int(ri.fallback) if ri is not None else 0,
                          rank=int(ri.rank) if ri is not None else -1,
                          p=int(ri.p) if ri is not None else int(idxs.sum()) * prob.kk, ties=ties)
    if ties["suspect"]:
        # the device solves the LASSO on Z^T Z, the reference on Z itself (Lasso.fit(Z, reY), decompose.py:449, 456): the two
        # agree to rounding, so a coefficient this close to the edge of its dead zone (or a stop this close to its
        # threshold) may have gone the other way there -- the mask is then not guaranteed identical
        warnings.warn("dictionary(): a LASSO decision was taken within a relative margin of %.1e of its threshold (edge margin "
                      "%s, duality-gap margin %s): the selected channels may differ from the CPU reference's at this tie"
                      % (TIE_MARGIN, ties["edge_margin"], ties["gap_margin"]), RuntimeWarning, stacklevel=2)
    if not dcfgs.autodet:
        cfgs.alpha = alpha_out                           # decompose.py:626-627 (`if not norank`)
    return idxs, newW2, newB2


def dictionary(X, W2, Y, alpha=1e-4, rank=None, DEBUG=0, B2=None, rank_tol=.1, verbose=0):
    """Channel selection by LASSO + least-squares reconstruction of W2 on the kept channels.
    (rank_tol is overridden by dcfgs.dic.rank_tol exactly as decompose.py:393 does.)"""
    _refit_mode()
    X = np.asarray(X)
    W2 = np.asarray(W2)
    if X.shape[2] != X.shape[-1]:
        raise ValueError("square kernels only (the reference assumes w = h, decompose.py:401-402)")
    # the operands are the caller's host arrays: only the sampled rows go to the device before the alpha search starts, X and
    # Y stream in behind it (cp_prune_layer_h2d); the other cd modes upload them first
    prob = LayerProblem(default_context(), X, W2, Y, flags=_flags(), defer_upload=True)
    try:
        idxs, newW2, newB2 = prune_resident(prob, rank, W2, alpha=alpha)
    finally:
        prob.free()
    if DEBUG:
        return X[:, idxs, ...], newW2, newB2             # decompose.py:629-632
    return idxs, newW2, newB2


class _FittedLinear:
    """What ``ret_reg=True`` callers use of the sklearn estimator (decompose.py:667-668, 681-685)."""

    def __init__(self, coef, intercept):
        self.coef_ = coef
        self.intercept_ = intercept

    def predict(self, X):
        return np.asarray(X) @ self.coef_.T + self.intercept_


def epscheck(x, tol=5):
    """decompose.py:158-161: warn when an entry exceeds 10**tol."""
    if np.any(np.abs(x) > 10 ** tol):
        print('1e' + str(tol) + ' exceed')


def VH_decompose(weights, rank=None, DEBUG=0, X=None, Y=None):
    """Spatial decomposition of a k x k convolution into (k x 1) then (1 x k) (decompose.py:85-146).

        weights [n, c, h, w] -> V [rank, c, h, 1], H [n, rank, 1, w], VHr [n, c, h, w] (and b when X, Y are given)

    The truncated SVD of the (c*h) x (n*w) matrix runs on the device (one-sided Jacobi, cp_svd_rows); with
    X[N, c, h, w] and Y[N, n] the second factor is refit by nonlinear_fc on Xv = X * V exactly as the reference
    does.  Singular vectors carry LAPACK's arbitrary sign in the reference and Jacobi's here: V[k] and H[:, k] may
    both be negated, their product (VHr) is the same."""
    weights = np.asarray(weights)
    dim = weights.shape
    ch, nw = dim[1] * dim[2], dim[0] * dim[3]
    VH = np.transpose(weights, [1, 2, 0, 3]).reshape([ch, nw])          # decompose.py:96-99
    if rank is None:
        rank = ch
    ctx = default_context()
    if ch <= nw:
        _, Vt, SH = ctx.svd_rows(VH, rank)                                # rows of Vt = V[:, k]; SH = diag(sigma) H
    else:                                                                 # work on the transpose, swap the roles
        sig, Ht, SV = ctx.svd_rows(VH.T, rank)
        Vt, SH = SV / sig[:, None], Ht * sig[:, None]
    V = Vt.T                                                              # ch x rank   (decompose.py:105-106)
    H = SH                                                                # rank x nw   (decompose.py:108-112)
    VHr = ctx.matmul_tn(Vt, H).reshape([dim[1], dim[2], dim[0], dim[3]])  # decompose.py:114
    H = H.reshape([rank, dim[0], dim[3], 1])                              # decompose.py:120-122
    H = np.transpose(H, [1, 0, 3, 2])
    origV = V.copy()
    V = V.reshape((dim[1], 1, dim[2], rank))                              # decompose.py:125-126
    V = np.transpose(V, [3, 0, 2, 1])
    b = None
    if X is not None:                                                     # decompose.py:128-139
        X = np.ascontiguousarray(X)
        if X.dtype not in (np.float32, np.float64):
            X = X.astype(np.float64)
        N = X.shape[0]
        o = H.shape[0]
        w = dim[3]
        Xd, Vd = ctx.to_device(X), ctx.to_device(np.ascontiguousarray(Vt))
        Xvd = ctx.empty(N * rank * w * 8)
        try:
            ctx._check(ctx.lib.cp_vh_project(ctx.h, Xd.ptr, _capi.CP_F32 if X.dtype == np.float32 else _capi.CP_F64, N,
                                             dim[1], dim[2], w, Vd.ptr, rank, Xvd.ptr), "cp_vh_project")
            Y2 = np.ascontiguousarray(Y, dtype=np.float64).reshape(N, -1)
            prob = LayerProblem.from_device(ctx, Xvd, _capi.CP_F64, N, rank * w, 1,
                                            np.zeros((Y2.shape[1], rank * w, 1, 1), dtype=np.float32), ctx.to_device(Y2))
            try:
                H2, b = prob.refit_nonlinear(np.ones(rank * w, dtype=bool))
            finally:
                prob.Yd.free()
                prob.free()
        finally:
            for bfr in (Xd, Vd, Xvd):
                bfr.free()
        H = H2.reshape([o, rank, 1, 3])                                   # decompose.py:137 (w = 3 hard-coded there)
        reH = np.transpose(H, [1, 0, 2, 3]).reshape([rank, -1])
        VHr = ctx.matmul_tn(Vt, reH).reshape([dim[1], dim[2], dim[0], dim[3]])
    VHr = np.transpose(VHr, [2, 0, 1, 3])                                 # decompose.py:141
    epscheck(V, 2)
    epscheck(H, 2)
    epscheck(VHr, 2)
    if X is not None:
        return V, H, VHr, b
    return V, H, VHr


def ITQ_decompose(feature, gt_feature, weight, rank, bias=None, DEBUG=False, Wr=None):
    """Channel decomposition of a convolution into rank filters followed by a 1 x 1 (decompose.py:163-319):

        feature, gt_feature [N, n]; weight [n, c, h, w] -> W1 [rank, c, h, w], W2 [n, rank, 1, 1], B [n], W12 [n, c, h, w]

    The 50 alternations (projection, rank-truncated SVD, ReLU-aware update) and the final SVD run on the device
    (cp_itq_iterate, cp_svd_rows); the weight products go through cp_matmul_tn."""
    feature = np.asarray(feature)
    gt_feature = np.asarray(gt_feature)
    n_ins = feature.shape[0]
    n_filter_channels = feature.shape[1]
    assert gt_feature.shape[0] == n_ins
    assert gt_feature.shape[1] == n_filter_channels
    ctx = default_context()
    T, Y_mean, U_mean = ctx.itq_iterate(feature, gt_feature, rank)        # decompose.py:170-246
    _, Lt, R = ctx.svd_rows(T, rank, lowrank=True)                        # decompose.py:249-252: L = Lt.T, R = diag(s) R
    L = np.ascontiguousarray(Lt.T)
    weight = np.asarray(weight)
    dim = weight.shape
    assert len(dim) == 4
    if dim[3] == n_filter_channels:
        assert False                                                      # decompose.py:286 (dead branch there too)
    assert dim[0] == n_filter_channels
    # right = 1 (decompose.py:258): W1 = weight^T-reshaped . L, i.e. (weight as [n, chw])^T L
    wt_shape = (dim[1], dim[2], dim[3])
    W1 = ctx.matmul_tn(np.ascontiguousarray(weight, dtype=np.float64).reshape(n_filter_channels, -1), L)   # [chw, rank]
    if Wr is not None:
        Wr = np.asarray(Wr)
        W12 = ctx.matmul_tn(np.ascontiguousarray(Wr, dtype=np.float64).reshape(n_filter_channels, -1), L)
        w12_shape = (Wr.shape[1], Wr.shape[2], Wr.shape[3])
    else:
        W12 = W1
        w12_shape = wt_shape
    W1 = np.transpose(W1.reshape(wt_shape + (rank,)), [3, 0, 1, 2])       # decompose.py:278-279
    W2 = R
    W12 = ctx.matmul_tn(np.ascontiguousarray(W12.T), W2)                  # decompose.py:293  [chw, n]
    W2 = W2.T.reshape([n_filter_channels, rank, 1, 1])                    # decompose.py:296-297
    W12 = np.transpose(W12.reshape(w12_shape + (n_filter_channels,)), [3, 0, 1, 2])
    B = -ctx.matmul_tn(Y_mean.reshape(-1, 1), T).reshape(-1) + U_mean      # decompose.py:305
    B = B.T + bias if bias is not None else B.T
    for arr in (W1, W2, B, W12):
        epscheck(arr, 2)
    for arr in (W1, W2, B, W12):
        epscheck(arr, 4)
    return W1, W2, B, W12


def nonlinear_fc(X, Y, copy_X=True, W=None, B=None):
    """ReLU-aware reconstruction (decompose.py:671-685): 30 + 20 alternations of fc_kernel and solve_relu with
    X[N,p] constant; returns (coef_[n, p], intercept_[n]) of the last regression."""
    assert len(X.shape) == 2
    assert copy_X == True  # noqa: E712
    assert W is None
    assert B is None
    X = np.ascontiguousarray(X)
    N, p = X.shape
    Y2 = np.ascontiguousarray(Y, dtype=np.float64).reshape(N, -1)
    prob = LayerProblem(default_context(), X.reshape(N, p, 1, 1), np.zeros((Y2.shape[1], p, 1, 1), dtype=np.float32), Y2)
    try:
        return prob.refit_nonlinear(np.ones(p, dtype=bool))
    finally:
        prob.free()


def fc_kernel(X, Y, copy_X=True, W=None, B=None, ret_reg=False, fit_intercept=True):
    """OLS with intercept (or Ridge when dcfgs.fc_ridge > 0) of Y[N,n] on X[N,p]: returns n x p."""
    assert copy_X == True  # noqa: E712  (decompose.py:640)
    assert len(X.shape) == 2
    if dcfgs.ls != 'linear':
        raise NotImplementedError("dcfgs.ls=%r needs modules that are not in the reference tree" % dcfgs.ls)
    if not fit_intercept:
        raise NotImplementedError("fit_intercept=False is never used on the pruning path")
    X = np.ascontiguousarray(X)
    N, p = X.shape
    Y2 = np.ascontiguousarray(Y, dtype=np.float64).reshape(N, -1)
    n = Y2.shape[1]
    ctx = default_context()
    # every column is a "channel" with a 1x1 kernel: the refit entry point then solves exactly this
    prob = LayerProblem(ctx, X.reshape(N, p, 1, 1), np.zeros((1, p, 1, 1), dtype=np.float32), np.zeros((N, 1)))
    try:
        prob.Yd.free()
        prob.Yd = ctx.to_device(Y2)
        prob.n = n
        prob.Wout.free()
        prob.bout.free()
        prob.Wout = ctx.empty(n * p * 8)
        prob.bout = ctx.empty(n * 8)
        coef, intercept = prob.refit(np.ones(p, dtype=bool), ridge=float(dcfgs.fc_ridge))
        ri = prob.refit_info
        last_call_info["refit_info"] = dict(p=int(ri.p), rank=int(ri.rank), fallback=int(ri.fallback))
    finally:
        prob.free()
    if np.ndim(Y) == 1:
        coef, intercept = coef[0], intercept[0]
    if ret_reg:
        return _FittedLinear(coef, intercept)
    return coef, intercept
