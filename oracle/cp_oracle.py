"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the channel-pruning hot path.

This module is the *checker*.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product
(``channel-pruning_amd/``) never does and fails loudly when its HIP library is missing.

What is restated, and from where (paths relative to /root/reference unless prefixed):

  dictionary_oracle()   lib/decompose.py:386-634   live path only (dcfgs defaults:
                        autodet=False, dic.alter=0, dic.debug=0, solver='sklearn',
                        ls='linear', nonlinear_fc=0, nofc=0; cfgs.py:70-111)
  fc_kernel_oracle()    lib/decompose.py:636-669   LinearRegression / Ridge branches
  rel_error()           lib/decompose.py:31-32
  lasso engines         'sklearn' : calls sklearn.linear_model.Lasso exactly as
                                    decompose.py:449,456-457 does (the reference's
                                    own third-party arithmetic, sklearn 1.7.2)
                        'c_data'  : oracle/cd_oracle.c cpo_enet_cd_data  (restates
                                    sklearn/linear_model/_cd_fast.pyx:101-273)
                        'c_gram'  : oracle/cd_oracle.c cpo_enet_cd_gram  (restates
                                    sklearn/linear_model/_cd_fast.pyx:564-737) -- the
                                    line-by-line spec of the HIP coordinate-descent kernel
  least squares         'sklearn' : LinearRegression -> scipy.linalg.lstsq(gelsd), as
                                    decompose.py:665-666
                        'numpy'   : lstsq_min_norm(): centring + truncated-SVD
                                    minimum-norm solution (gelsd's published algorithm;
                                    cut-off sigma_i <= max(N,p)*eps*sigma_max,
                                    sklearn/linear_model/_base.py:700-701)

Third-party versions the oracle (and every golden file) is pinned to:
numpy 2.2.6, scipy 1.15.3, scikit-learn 1.7.2 (all present here and on the GPU box).

Pinning: the reference ships no tests for this path, so the
oracle is pinned against outputs of the *unmodified* reference function run in the build
container (oracle/gen_golden.py -> tests/golden/*.npz; oracle/validate_oracle.py and
tests/test_oracle.py compare).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcporacle.so")
_lib = None

RAND_R_MAX = 2147483647  # sklearn/linear_model/_cd_fast.pyx:26


def build(force=False):
    """Compile oracle/cd_oracle.c (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "cd_oracle.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def _c():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.cpo_enet_cd_data.restype = ctypes.c_int
        _lib.cpo_enet_cd_data.argtypes = [dp, ctypes.c_double, ctypes.c_double, dp, dp,
                                          ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_double, ctypes.c_uint32, ctypes.c_int32, dp, dp]
        _lib.cpo_enet_cd_gram.restype = ctypes.c_int
        _lib.cpo_enet_cd_gram.argtypes = [dp, ctypes.c_double, ctypes.c_double, dp, dp,
                                          ctypes.c_double, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_double, ctypes.c_uint32, ctypes.c_int32,
                                          ctypes.c_int32, dp]
        _lib.cpo_coord_sequence.restype = None
        _lib.cpo_coord_sequence.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int64,
                                            ctypes.POINTER(ctypes.c_int32)]
        _lib.cpo_lasso_operands.restype = None
        _lib.cpo_lasso_operands.argtypes = [dp, dp, dp, ctypes.POINTER(ctypes.c_int64),
                                            ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, dp, dp, dp, dp, dp, dp]
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        _lib.cpo_patch_gather.restype = None
        _lib.cpo_patch_gather.argtypes = [fp] + [ctypes.c_int32] * 4 + [ip, ip] + \
            [ctypes.c_int32] * 5 + [fp]
        _lib.cpo_assemble_y.restype = None
        _lib.cpo_assemble_y.argtypes = [fp, fp, dp, ctypes.c_int64, ctypes.c_int32, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


# --------------------------------------------------------------------------------------
# kernel-level restatements (C)
# --------------------------------------------------------------------------------------
def coord_sequence(seed, n_features, count):
    out = np.empty(count, dtype=np.int32)
    _c().cpo_coord_sequence(int(seed), int(n_features), int(count),
                            out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out


def enet_cd_data(w, l1_reg, l2_reg, Xc, yc, max_iter=1000, tol=1e-4, seed=1, random=True):
    """sklearn/_cd_fast.pyx:101-273.  Xc Fortran-ordered centred [M, c]; w updated in place."""
    Xc = np.asfortranarray(Xc, dtype=np.float64)
    yc = np.ascontiguousarray(yc, dtype=np.float64)
    assert w.dtype == np.float64 and w.flags.c_contiguous
    gap = ctypes.c_double()
    tols = ctypes.c_double()
    n_iter = _c().cpo_enet_cd_data(_dp(w), l1_reg, l2_reg, _dp(Xc), _dp(yc), Xc.shape[0],
                                   Xc.shape[1], max_iter, tol, int(seed), int(random),
                                   ctypes.byref(gap), ctypes.byref(tols))
    return w, gap.value, tols.value, n_iter


def enet_cd_gram(w, l1_reg, l2_reg, Q, q, y_norm2, max_iter=1000, tol=1e-4, seed=1,
                 random=True, recip=False, delta=False):
    """sklearn/_cd_fast.pyx:564-737.  Returns (w, stats[gap,tol,q.w,|XtA|inf,R2], n_iter).
    recip / delta: the device kernel's CP_CD_RECIPROCAL / CP_CD_DELTA rounding variants."""
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    assert w.dtype == np.float64 and w.flags.c_contiguous
    stats = np.zeros(5)
    n_iter = _c().cpo_enet_cd_gram(_dp(w), l1_reg, l2_reg, _dp(Q), _dp(q), float(y_norm2),
                                   Q.shape[0], max_iter, tol, int(seed), int(random),
                                   int(bool(recip)) | (2 if delta else 0), _dp(stats))
    return w, stats, n_iter


def lasso_operands(X, W2, Y, samples, want_Z=True):
    """decompose.py:425-437 + sklearn centring.  X[N,c,k,k], W2[n,c,k,k], Y[N,n]."""
    N, c = X.shape[0], X.shape[1]
    n = W2.shape[0]
    kk = int(np.prod(X.shape[2:]))
    X = np.ascontiguousarray(X, dtype=np.float64).reshape(N, c, kk)
    W2 = np.ascontiguousarray(W2, dtype=np.float64).reshape(n, c, kk)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    samples = np.ascontiguousarray(samples, dtype=np.int64)
    S = samples.shape[0]
    M = S * n
    Zc = np.empty((M, c), dtype=np.float64, order="F") if want_Z else None
    yc = np.empty(M, dtype=np.float64)
    Q = np.empty((c, c))
    q = np.empty(c)
    stats = np.empty(2)
    zmean = np.empty(c)
    _c().cpo_lasso_operands(_dp(X), _dp(W2), _dp(Y),
                            samples.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), S, c, n, kk,
                            _dp(Zc) if want_Z else None, _dp(yc), _dp(Q), _dp(q), _dp(stats),
                            _dp(zmean))
    return dict(Zc=Zc, yc=yc, Q=Q, q=q, yty=stats[0], ymean=stats[1], zmean=zmean, M=M)


def patch_gather(fmap, xs, ys, k, pad, stride, relu):
    """net.py:629-657 for one batch.  fmap [B,C,H,W] f32 -> [P*B, C, k, k] f32."""
    fmap = np.ascontiguousarray(fmap, dtype=np.float32)
    B, C, H, W = fmap.shape
    xs = np.ascontiguousarray(xs, dtype=np.int32)
    ys = np.ascontiguousarray(ys, dtype=np.int32)
    P = xs.shape[0]
    out = np.empty((P * B, C, k, k), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int32)
    _c().cpo_patch_gather(fmap.ctypes.data_as(fp), B, C, H, W, xs.ctypes.data_as(ip),
                          ys.ctypes.data_as(ip), P, k, pad, stride, int(relu),
                          out.ctypes.data_as(fp))
    return out


def assemble_y(feats, bias, resY=None):
    """net.py:1707,1722: Y = feats - bias (+ resY) in float64 from float32 blobs."""
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    bias = np.ascontiguousarray(bias, dtype=np.float32)
    N, n = feats.shape
    Y = np.empty((N, n))
    fp = ctypes.POINTER(ctypes.c_float)
    r = None
    if resY is not None:
        resY = np.ascontiguousarray(resY, dtype=np.float64)
        r = _dp(resY)
    _c().cpo_assemble_y(feats.ctypes.data_as(fp), bias.ctypes.data_as(fp), r, N, n, _dp(Y))
    return Y


# --------------------------------------------------------------------------------------
# host-level restatements (numpy)
# --------------------------------------------------------------------------------------
def rel_error(A, B):
    """decompose.py:31-32"""
    return np.mean((A - B) ** 2) ** .5 / np.mean(A ** 2) ** .5


def lstsq_min_norm(X, Y, ridge=0.0):
    """OLS with intercept as LinearRegression.fit does it (_base.py:591-706): centre X and Y
    by their column means, minimum-norm least squares with singular values
    sigma_i <= max(N,p)*eps*sigma_max dropped (gelsd), intercept = ybar - xbar.coef^T
    (_base.py:300-316).  ridge>0: (Xc^T Xc + ridge I)^-1 Xc^T Yc (sklearn Ridge, dense)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    xm = X.mean(axis=0)
    ym = Y.mean(axis=0)
    Xc = X - xm
    Yc = Y - ym
    if ridge > 0:
        G = Xc.T @ Xc
        G[np.diag_indices_from(G)] += ridge
        coef = np.linalg.solve(G, Xc.T @ Yc).T
        rank = X.shape[1]
    else:
        U, s, Vt = np.linalg.svd(Xc, full_matrices=False)
        cond = max(X.shape) * np.finfo(np.float64).eps
        rank = int(np.sum(s > cond * s[0])) if s.size and s[0] > 0 else 0
        coef = ((Vt[:rank].T / s[:rank]) @ (U[:, :rank].T @ Yc)).T
    intercept = ym - xm @ coef.T
    return coef, intercept, rank


def fc_kernel_oracle(X, Y, ridge=0.0, engine="sklearn"):
    """decompose.py:636-669 (LinearRegression / Ridge branches) -> (coef_[n,p], intercept_[n])."""
    assert X.ndim == 2
    if engine == "numpy":
        coef, b, _ = lstsq_min_norm(X, Y, ridge)
        return coef, b
    from sklearn.linear_model import LinearRegression, Ridge
    reg = Ridge(alpha=ridge) if ridge > 0 else LinearRegression(n_jobs=-1, copy_X=True,
                                                               fit_intercept=True)
    reg.fit(X, Y)
    return reg.coef_, reg.intercept_


class _LassoEngine:
    """solve(alpha) of decompose.py:453-466 for the three engines; keeps the warm start."""

    def __init__(self, Z, reY, engine, rng):
        self.engine = engine
        self.rng = rng
        self.M, self.c = Z.shape
        if engine == "sklearn":
            from sklearn.linear_model import Lasso
            self.Z, self.reY = Z, reY
            # decompose.py:449; random_state=None uses numpy's global RandomState, which
            # is what `rng` is unless the caller supplied its own stream.
            self.solver = Lasso(alpha=1e-4, warm_start=True, selection="random",
                                random_state=None if rng is np.random else rng)
        else:
            # _pre_fit -> _preprocess_data: centre columns of Z and y (fit_intercept=True)
            self.Zc = np.asfortranarray(Z - Z.mean(axis=0))
            self.yc = reY - reY.mean()
            self.w = np.zeros(self.c)
            if engine == "c_gram":
                self.Q = np.ascontiguousarray(self.Zc.T @ self.Zc)
                self.q = self.Zc.T @ self.yc
                self.yty = float(self.yc @ self.yc)

    def solve(self, alpha):
        if self.engine == "sklearn":
            self.solver.alpha = alpha
            self.solver.fit(self.Z, self.reY)
            coef, n_iter, seed = self.solver.coef_, self.solver.n_iter_, None
        else:
            seed = self.rng.randint(0, RAND_R_MAX)          # _cd_fast.pyx:164 / 626
            l1 = alpha * self.M                              # _coordinate_descent.py:653
            if self.engine == "c_data":
                _, _, _, n_iter = enet_cd_data(self.w, l1, 0.0, self.Zc, self.yc, 1000, 1e-4, seed)
            else:
                _, _, n_iter = enet_cd_gram(self.w, l1, 0.0, self.Q, self.q, self.yty, 1000,
                                            1e-4, seed)
            coef = self.w
        idxs = coef != 0.
        return idxs, int(idxs.sum()), int(n_iter), seed


def solve_relu_oracle(RU, Z, Lambda):
    """decompose.py:51-59, statement by statement."""
    U0 = np.minimum(RU, 0.)
    Cost0 = Z ** 2 + Lambda * (U0 - RU) ** 2
    U1 = np.maximum((Lambda * RU + Z) / (Lambda + 1.), 0.)
    Cost1 = (U1 - Z) ** 2 + Lambda * (U1 - RU) ** 2
    return (Cost0 <= Cost1) * U0 + (Cost0 > Cost1) * U1


def nonlinear_fc_oracle(X, Y, engine="sklearn"):
    """decompose.py:671-685: U = Y, Z = relu(Y); 30 iterations with lambda = 0.1 then 20 with lambda = 1 of
    reg = fc_kernel(X, U); U = solve_relu(reg.predict(X), Z, lambda); returns the last reg's (coef_, intercept_)."""
    assert X.ndim == 2
    U = Y.copy()
    Z = np.maximum(Y, 0.)
    its = [30, 20]
    coef = b = None
    for epoch, l in enumerate([10 ** i for i in range(-1, 1)]):
        for _ in range(its[epoch]):
            coef, b = fc_kernel_oracle(X, U, engine=engine)
            RU = X @ coef.T + b
            U = solve_relu_oracle(RU, Z, l)
    return coef, b


def vh_decompose_oracle(weights, rank=None, X=None, Y=None, engine="sklearn"):
    """decompose.py:85-146 restated (scipy gesvd as the reference calls it): V[rank,c,h,1], H[n,rank,1,w],
    VHr[n,c,h,w] (+ b with X, Y).  Singular-vector signs are LAPACK's."""
    import scipy.linalg
    dim = weights.shape
    VH = np.transpose(weights, [1, 2, 0, 3]).reshape([dim[1] * dim[2], dim[0] * dim[3]])
    V, s, H = scipy.linalg.svd(VH, full_matrices=False, lapack_driver='gesvd')
    if rank is None:
        rank = dim[1] * dim[2]
    V = V[:, :rank]
    H = np.diag(s[:rank]).dot(H[:rank, :])
    VHr = (V.dot(H)).reshape([dim[1], dim[2], dim[0], dim[3]])
    H = np.transpose(H.reshape([rank, dim[0], dim[3], 1]), [1, 0, 3, 2])
    origV = V.copy()
    V = np.transpose(V.reshape((dim[1], 1, dim[2], rank)), [3, 0, 2, 1])
    b = None
    if X is not None:
        Xv = np.transpose(np.tensordot(X, V, [[1, 2], [1, 2]]), [0, 2, 3, 1])
        N, o = Xv.shape[0], H.shape[0]
        H, b = nonlinear_fc_oracle(Xv.reshape([N, -1]), Y, engine=engine)
        H = H.reshape([o, rank, 1, 3])
        reH = np.transpose(H, [1, 0, 2, 3]).reshape([rank, -1])
        VHr = (origV.dot(reH)).reshape([dim[1], dim[2], dim[0], dim[3]])
    VHr = np.transpose(VHr, [2, 0, 1, 3])
    return (V, H, VHr, b) if X is not None else (V, H, VHr)


def itq_decompose_oracle(feature, gt_feature, weight, rank, bias=None, Wr=None):
    """decompose.py:163-319 restated (scipy gesvd, pinv with the 1e-6 relative cut-off): W1, W2, B, W12."""
    import scipy.linalg
    svd = lambda x: scipy.linalg.svd(x, full_matrices=False, lapack_driver='gesvd')  # noqa: E731
    n = feature.shape[1]
    Y = feature.copy()
    Z = np.maximum(gt_feature, 0.)
    Zsq = Z ** 2
    Y_mean = Y.mean(0)
    G = Y - Y_mean
    PG = scipy.linalg.pinv((G.T).dot(G), rtol=1e-6)
    PGGt = PG.dot(G.T)
    UU = G.copy()
    U_mean = Y_mean.copy()
    T = None
    for Lambda, its in zip([0.1, 1], [30, 20]):
        for _ in range(its):
            X = G.dot(PGGt.dot(UU))
            L, sigma, R = svd(X)
            T = L[:, :rank].dot(np.diag(sigma[:rank])).dot(R[:rank, :])
            T = PGGt.dot(T)
            RU = G.dot(T)
            RU += U_mean
            U0 = np.minimum(RU, 0.)
            Cost0 = Zsq + Lambda * (U0 - RU) ** 2
            U1 = np.maximum((Lambda * RU + Z) / (Lambda + 1.), 0.)
            Cost1 = (U1 - Z) ** 2 + Lambda * (U1 - RU) ** 2
            U = (Cost0 <= Cost1) * U0 + (Cost0 > Cost1) * U1
            U_mean = U.mean(0)
            UU = U - U_mean
    L, sigma, R = svd(T)
    L = L[:, :rank]
    R = np.diag(sigma[:rank]).dot(R[:rank, :])
    assert weight.shape[0] == n and weight.shape[3] != n
    wt = np.transpose(weight, [1, 2, 3, 0])
    W1 = wt.reshape([-1, n]).dot(L)
    if Wr is not None:
        Wrt = np.transpose(Wr, [1, 2, 3, 0])
        W12 = Wrt.reshape([-1, n]).dot(L)
        shp12 = Wrt.shape[:3]
    else:
        W12 = W1
        shp12 = wt.shape[:3]
    W1 = np.transpose(W1.reshape(wt.shape[:3] + (rank,)), [3, 0, 1, 2])
    W2 = R
    W12 = W12.dot(W2)
    W2 = W2.T.reshape([n, rank, 1, 1])
    W12 = np.transpose(W12.reshape(shp12 + (n,)), [3, 0, 1, 2])
    B = -Y_mean.dot(T) + U_mean
    B = B.T + bias if bias is not None else B.T
    return W1, W2, B, W12


def dictionary_oracle(X, W2, Y, rank, B2=None, alpha=1e-4, alpha_in=1e-3, rank_tol=.1, rng=None,
                      lasso="sklearn", ls="sklearn", ridge=0.0, log=None, refit="linear", autodet=False, layeralpha=1):
    """Restatement of lib/decompose.py:386-634 (live path).

    alpha_in plays the role of the module global ``cfgs.alpha`` on entry (decompose.py:491);
    the value the reference writes back (decompose.py:626-627) is returned as the 4th item.
    rng: numpy global RNG by default (decompose.py:425; the reference never seeds it).
    Returns (idxs bool[c], newW2 f64[n, nnz, k, k], newB2 f64[n], alpha_out)."""
    rng = np.random if rng is None else rng
    N, c, h = X.shape[0], X.shape[1], X.shape[2]
    w = h                                                     # decompose.py:401-402
    n = W2.shape[0]
    samples = rng.randint(0, N, min(400, N // 20))            # decompose.py:425
    reX = np.rollaxis(X.reshape((N, c, -1))[samples], 1, 0)   # c S hw
    reW2 = np.transpose(W2.reshape((n, c, -1)), [1, 2, 0])    # c hw n
    Z = np.matmul(reX, reW2).reshape((c, -1)).T               # [S*n, c]
    reY = Y[samples].reshape(-1)
    if log is not None:
        log.append(("samples", samples.copy()))
    if autodet:                                               # decompose.py:395-397, 414-415, 582-585
        alpha = alpha_in / c ** layeralpha
        eng = _LassoEngine(Z, reY, lasso, rng)
        idxs, rank, n_iter, seed = eng.solve(alpha)
        if log is not None:
            log.append(("fit", alpha, rank, n_iter, seed))
    elif rank == c:                                           # decompose.py:487-488
        idxs = np.array([True] * rank)
    else:
        eng = _LassoEngine(Z, reY, lasso, rng)
        left, right = 0, alpha_in                             # decompose.py:490-491
        lbound = rank
        if rank_tol >= 1:
            rbound = rank + rank_tol
        else:
            rbound = rank + rank_tol * rank
            if rank_tol == .2:                                # decompose.py:498-501
                lbound = rank + 0.1 * rank
                rbound = rank + 0.2 * rank
        while True:                                           # decompose.py:502-515
            _, tmp, n_iter, seed = eng.solve(right)
            if log is not None:
                log.append(("fit", right, tmp, n_iter, seed))
            if tmp < rank:
                break
            right *= 2
        while True:                                           # decompose.py:516-525
            alpha = (left + right) / 2
            idxs, tmp, n_iter, seed = eng.solve(alpha)
            if log is not None:
                log.append(("fit", alpha, tmp, n_iter, seed))
            if tmp > rbound:
                left = alpha
            elif tmp < lbound:
                right = alpha
            else:
                break
        rank = tmp                                            # decompose.py:581
    if refit == "nonlinear":                                  # decompose.py:615-617
        newW2, newB2 = nonlinear_fc_oracle(X[:, idxs, ...].reshape((N, -1)), Y, engine=ls)
    elif refit == "none":                                     # decompose.py:618-620
        return idxs, W2[:, idxs, :, :], np.zeros(n), alpha
    else:
        newW2, newB2 = fc_kernel_oracle(X[:, idxs, ...].reshape((N, -1)), Y, ridge=ridge, engine=ls)
    newW2 = newW2.reshape((n, rank, h, w))                    # decompose.py:622-623
    if autodet:
        return idxs, newW2, newB2, alpha_in                   # `if not norank: cfgs.alpha = alpha` (decompose.py:626)
    return idxs, newW2, newB2, alpha                          # decompose.py:626-627, 634


SKETCH_COLS = 32


def sketch_matrix(p):
    """Seeded Gaussian test matrix Omega[p, 32] of the sketched weight goldens (tests/golden/V*.npz, L13):
    ||(W - Wref) Omega||_F / ||Wref Omega||_F estimates the relative Frobenius error of W (E[||A Omega||^2] = 32 ||A||^2)."""
    return np.random.RandomState(777).randn(int(p), SKETCH_COLS)


def mix_channels(X, rs, mix):
    """Ill-conditioned channel structure for the q* goldens (float64 in, float64 out; X[N,c,k,k]):
       kappa   : channels mixed by a c x c matrix with log-spaced singular values 1 .. 1/kappa
       dup     : three channels are copies of three others + eps * noise (eps = 0: exact copies, rank-deficient)
       relumix : relu of a rank-r mixture of latent maps + delta of its own full-rank remainder
       powerlaw: relu of a full-rank mixture of latent maps whose singular values fall off like i^-p -- strongly
                 correlated channels with a long spectrum, the shape of real post-ReLU activations (unlike i.i.d. columns)"""
    N, c = X.shape[0], X.shape[1]
    kind = mix["kind"]
    if kind == "kappa":
        U, _ = np.linalg.qr(rs.randn(c, c))
        V, _ = np.linalg.qr(rs.randn(c, c))
        T = (U * np.logspace(0, -np.log10(mix["kappa"]), c)) @ V.T
        return np.einsum("nikl,ij->njkl", X, T) * np.sqrt(c)
    if kind == "powerlaw":
        U, _ = np.linalg.qr(rs.randn(c, c))
        V, _ = np.linalg.qr(rs.randn(c, c))
        T = (U * np.arange(1, c + 1, dtype=np.float64) ** -mix["p"]) @ V.T
        G = rs.randn(*X.shape)
        Z = np.einsum("nikl,ij->njkl", G, T)
        return np.maximum(Z / Z.std(), 0.)
    if kind == "dup":
        X = X.copy()
        for (i, j) in ((1, 9), (4, 5), (20, 2)):
            X[:, j] = X[:, i] + mix["eps"] * rs.randn(*X[:, i].shape)
        return X
    if kind == "relumix":
        r = mix["r"]
        G = rs.randn(N, r, X.shape[2], X.shape[3])
        A = rs.randn(r, c) / np.sqrt(r)
        low = np.maximum(np.einsum("nrkl,rc->nckl", G, A), 0.)
        flat = low.reshape(N, -1)
        fc = flat - flat.mean(0)
        Uu, ss, Vt = np.linalg.svd(fc, full_matrices=False)
        rr = r * X.shape[2] * X.shape[3]
        lo = (Uu[:, :rr] * ss[:rr]) @ Vt[:rr]
        return (flat.mean(0) + lo + mix["delta"] * (fc - lo)).reshape(X.shape)
    raise ValueError(kind)


def synth_layer(layer_id, N, c, n, k, relu=True, dtype=np.float32, noise=0.01, dead=0,
                residual=False, mix=None):
    """Synthetic operands of SURVEY.md section 8d / BASELINE.md section 2 (the generator the
    CPU probes used): seeds RandomState(1000+layer_id).  dead>0 zeroes that many channels
    of X (ReLU-dead channels); residual adds a dense extra term to Y and skips the ReLU
    (ResNet case, net.py:1716-1722)."""
    rs = np.random.RandomState(1000 + layer_id)
    X = rs.randn(N, c, k, k)
    if relu and not residual:
        X = np.maximum(X, 0.)
    X = X.astype(dtype)
    if mix is not None:
        X = mix_channels(X.astype(np.float64), rs, mix)
    if dead:
        X[:, rs.choice(c, dead, replace=False)] = 0
    W2 = (rs.randn(n, c, k, k) * 0.05).astype(np.float32)
    B2 = np.zeros(n, dtype=np.float32)
    Y = X.reshape(N, -1).astype(np.float64) @ W2.reshape(n, -1).T.astype(np.float64) \
        + noise * rs.randn(N, n)
    if residual:
        Y = Y + 0.1 * rs.randn(N, n)
    return X, W2, Y, B2
