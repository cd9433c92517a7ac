"""CPU: pins the oracle (test infrastructure) against
  (a) the golden vectors produced by the UNMODIFIED reference (tests/golden/, oracle/gen_golden.py),
  (b) the third-party code whose algorithm it restates (scikit-learn's Cython CD, scipy's gelsd),
  (c) plain numpy restatements of the reference's operand construction.
No GPU, no product code."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

import cp_oracle
from conftest import GOLDEN_DIR, ROOT


def relfro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / nb if nb > 0 else np.linalg.norm(a - b)


SMALL = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "s*.npz")))


def run_oracle(name, lasso, ls):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = json.loads(str(g["params"]))
    X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"], dead=p.get("dead", 0),
                                         residual=p.get("residual", False))
    np.random.seed(1234 + p["layer_id"])
    log = []
    idxs, newW2, newB2, alpha_out = cp_oracle.dictionary_oracle(
        X.astype(np.float64), W2, Y, p["rank"], B2, alpha_in=p.get("alpha_in", 1e-3),
        rank_tol=p.get("rank_tol", .1), lasso=lasso, ls=ls, ridge=p.get("fc_ridge", 0.0), log=log,
        refit="nonlinear" if p.get("nonlinear_fc") else ("none" if p.get("nofc") else "linear"),
        autodet=bool(p.get("autodet", 0)))
    rng_next = int(np.random.randint(0, 2147483647))
    fits = np.array([(f[1], f[2], f[3]) for f in log if f[0] == "fit"], dtype=np.float64).reshape(-1, 3)
    return g, idxs, newW2, newB2, alpha_out, rng_next, fits


def test_golden_files_present_and_versioned():
    import scipy
    import sklearn
    files = glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
    assert len(files) >= 18
    v = json.loads(str(np.load(files[0])["versions"]))
    # the oracle is "reference code + these third-party versions" 
    assert v["sklearn"] == sklearn.__version__ and v["scipy"] == scipy.__version__ and v["numpy"] == np.__version__


@pytest.mark.parametrize("name", SMALL)
def test_oracle_sklearn_engine_reproduces_reference_bitwise(name):
    g, idxs, newW2, newB2, alpha_out, rng_next, fits = run_oracle(name, "sklearn", "sklearn")
    assert np.array_equal(idxs, g["idxs"])
    assert np.array_equal(fits, g["fits"])
    assert np.array_equal(newW2, g["newW2"]) and np.array_equal(newB2, g["newB2"])
    assert alpha_out == float(g["alpha_out"]) and rng_next == int(g["rng_next"])


@pytest.mark.parametrize("name", SMALL + ["m01_config1"])
@pytest.mark.parametrize("engine", ["c_data", "c_gram"])
def test_oracle_c_engines_reproduce_reference(name, engine):
    """C restatements of sklearn's CD (data form = what the reference runs; Gram form = the spec of
    the HIP kernel): identical masks, per-fit (alpha, nnz, n_iter), RNG consumption; weights 1e-9."""
    g, idxs, newW2, newB2, alpha_out, rng_next, fits = run_oracle(name, engine, "numpy")
    assert np.array_equal(idxs, g["idxs"])
    assert np.array_equal(fits, g["fits"])
    assert alpha_out == float(g["alpha_out"]) and rng_next == int(g["rng_next"])
    assert relfro(newW2, g["newW2"]) <= 1e-9 and relfro(newB2, g["newB2"]) <= 1e-9


def _lasso_problem(M=3000, c=40, seed=0):
    rs = np.random.RandomState(seed)
    Z = rs.randn(M, c) * (0.3 + rs.rand(c))
    w = np.where(rs.rand(c) < 0.5, rs.randn(c), 0)
    y = Z @ w + 0.05 * rs.randn(M) + 2.0
    return Z, y


def test_c_data_form_matches_sklearn_lasso():
    """Same seed stream -> same n_iter and zero pattern as sklearn.linear_model.Lasso itself."""
    from sklearn.linear_model import Lasso
    Z, y = _lasso_problem()
    M = Z.shape[0]
    Zc = np.asfortranarray(Z - Z.mean(0))
    yc = y - y.mean()
    w = np.zeros(Z.shape[1])
    solver = Lasso(alpha=0.1, warm_start=True, selection="random")
    for i, alpha in enumerate([0.2, 0.05, 0.1, 0.01]):
        np.random.seed(100 + i)
        solver.alpha = alpha
        solver.fit(Z, y)
        np.random.seed(100 + i)
        seed = np.random.randint(0, cp_oracle.RAND_R_MAX)
        _, gap, tol, n_iter = cp_oracle.enet_cd_data(w, alpha * M, 0.0, Zc, yc, 1000, 1e-4, seed)
        assert n_iter == solver.n_iter_
        assert np.array_equal(w != 0, solver.coef_ != 0)
        assert relfro(w, solver.coef_) <= 1e-10


def test_c_gram_form_matches_sklearn_gram_kernel():
    """cpo_enet_cd_gram vs sklearn's enet_coordinate_descent_gram (same Q, q, rng): same n_iter,
    same zero pattern, coefficients to 1e-12 (BLAS vs explicit-fma rounding only)."""
    from sklearn.linear_model import _cd_fast
    Z, y = _lasso_problem(seed=1)
    Zc = Z - Z.mean(0)
    yc = y - y.mean()
    Q = np.ascontiguousarray(Zc.T @ Zc)
    q = Zc.T @ yc
    yty = float(yc @ yc)
    M = Z.shape[0]
    w_sk = np.zeros(Z.shape[1])
    w_c = np.zeros(Z.shape[1])
    for i, alpha in enumerate([0.2, 0.05, 0.1]):
        rng = np.random.RandomState(7 + i)
        out = _cd_fast.enet_coordinate_descent_gram(w_sk, alpha * M, 0.0, Q, q, yc, 1000, 1e-4, rng, True, False)
        seed = np.random.RandomState(7 + i).randint(0, cp_oracle.RAND_R_MAX)
        _, stats, n_iter = cp_oracle.enet_cd_gram(w_c, alpha * M, 0.0, Q, q, yty, 1000, 1e-4, seed)
        assert n_iter == out[3]
        assert np.array_equal(w_c != 0, np.asarray(out[0]) != 0)
        assert relfro(w_c, out[0]) <= 1e-12
        assert abs(stats[0] - out[1]) <= 1e-8 * max(1.0, abs(out[2]))


def test_rand_r_sequence_matches_published_xorshift():
    """our_rand_r (sklearn/utils/_random.pxd:20-35): x ^= x<<13; x ^= x>>17; x ^= x<<5; % 2^31; then % n."""
    def ref(seed, n, count):
        s = np.uint32(seed if seed else 1)
        out = []
        for _ in range(count):
            s ^= np.uint32(s << np.uint32(13))
            s ^= np.uint32(s >> np.uint32(17))
            s ^= np.uint32(s << np.uint32(5))
            out.append(int(s % np.uint32(2147483648)) % n)
        return np.array(out, dtype=np.int32)
    with np.errstate(over="ignore"):
        for seed, n in ((1, 256), (0, 7), (2147483646, 512), (12345, 55)):
            assert np.array_equal(cp_oracle.coord_sequence(seed, n, 500), ref(seed, n, 500))


def test_persistent_cholesky_task_order_is_a_topological_order_host_build():
    """tests/host/test_chain_order.cpp walks the linear task order of the persistent blocked Cholesky (csrc/chain_order.h, the
    header k_chol_chain itself decodes its tasks with) for 1154 shapes (1 .. 48 block rows, 0 .. 32 right-hand-side tile columns,
    lazy periods 1 .. 4, and 144 block rows): every task finds its tile at exactly the version it expects and every finished tile
    it multiplies with ALREADY final -- i.e. handing the tasks out off one counter cannot deadlock --, every tile receives
    every block row once, in order, and ends final."""
    exe = "/tmp/cp_test_chain_order"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "channel-pruning_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "test_chain_order.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "chain order ok" in out.stdout, out.stdout[-2000:]


def test_xorshift_jump_ahead_host_build():
    """The device kernel's batched index stream (xorshift_jump.h) checked against the sequential
    generator by a g++-compiled host program."""
    exe = "/tmp/cp_test_xorshift_jump"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "channel-pruning_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "test_xorshift_jump.cpp"), "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert out.startswith("ok")


def test_lasso_operands_match_numpy():
    """decompose.py:425-437 written with numpy, against the C restatement."""
    rs = np.random.RandomState(3)
    N, c, n, k = 300, 24, 20, 3
    X = np.maximum(rs.randn(N, c, k, k), 0)
    W2 = rs.randn(n, c, k, k) * 0.1
    Y = rs.randn(N, n)
    samples = rs.randint(0, N, 15)
    reX = np.rollaxis(X.reshape((N, c, -1))[samples], 1, 0)
    reW2 = np.transpose(W2.reshape((n, c, -1)), [1, 2, 0])
    Z = np.matmul(reX, reW2).reshape((c, -1)).T
    reY = Y[samples].reshape(-1)
    Zc = Z - Z.mean(0)
    yc = reY - reY.mean()
    o = cp_oracle.lasso_operands(X, W2, Y, samples)
    assert relfro(o["Zc"], Zc) <= 1e-13 and relfro(o["yc"], yc) <= 1e-13
    assert relfro(o["Q"], Zc.T @ Zc) <= 1e-13 and relfro(o["q"], Zc.T @ yc) <= 1e-12
    assert abs(o["yty"] - yc @ yc) <= 1e-12 * (yc @ yc)


def test_patch_gather_matches_reference_loop():
    """extract_XY's window copy (net.py:629-657) + reshape/rollaxis (net.py:1702), in numpy."""
    rs = np.random.RandomState(4)
    B, C, H, W, k, P = 3, 5, 8, 8, 3, 6
    for pad, stride in ((1, 1), (0, 1), (1, 2)):
        fmap = rs.randn(B, C, H, W).astype(np.float32)
        top = (H + 2 * pad - k) // stride + 1
        xs, ys = rs.randint(0, top, P), rs.randint(0, top, P)
        feat = np.zeros((B, C, H + 2 * pad, W + 2 * pad), dtype=np.float32)
        feat[:, :, pad:H + pad, pad:W + pad] = fmap
        hk = k // 2
        rows = np.ndarray((P * B * k * k, C))
        for point, (x, y) in enumerate(zip(xs, ys)):
            x0, y0 = hk + stride * x, hk + stride * y           # top2bottom (padded), net.py:566-572
            win = feat[:, :, x0 - hk:x0 + hk + 1, y0 - hk:y0 + hk + 1]
            rows[point * B * k * k:(point + 1) * B * k * k] = np.moveaxis(win, 1, -1).reshape((B * k * k, -1))
        newX = np.rollaxis(rows.reshape((-1, k, k, C)), 3, 1)
        got = cp_oracle.patch_gather(fmap, xs, ys, k, pad, stride, relu=0)
        assert np.array_equal(got.astype(np.float64), newX)
        assert np.array_equal(cp_oracle.patch_gather(fmap, xs, ys, k, pad, stride, relu=1), np.maximum(got, 0))


def test_lstsq_min_norm_matches_scipy_gelsd():
    from scipy import linalg
    rs = np.random.RandomState(5)
    for N, p in ((300, 40), (50, 80)):
        X = rs.randn(N, p)
        X[:, 3] = 0                                        # a dead column
        Y = rs.randn(N, 7)
        coef, b, rank = cp_oracle.lstsq_min_norm(X, Y)
        Xc, Yc = X - X.mean(0), Y - Y.mean(0)
        ref, _, rk, _ = linalg.lstsq(Xc, Yc, cond=max(X.shape) * np.finfo(float).eps)
        assert rank == rk
        assert relfro(coef, ref.T) <= 1e-10
        assert relfro(b, Y.mean(0) - X.mean(0) @ ref) <= 1e-10
        assert np.abs(coef[:, 3]).max() <= 1e-13   # dead column: (numerically) zero weight


# ---- VH_decompose (decompose.py:85-146) ---------------------------------------------------------------
def _vh_case(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = json.loads(str(g["params"]))
    X, W2, Y, _ = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"])
    return g, p, X.astype(np.float64), W2.astype(np.float64), Y


def test_vh_oracle_reproduces_reference_svd_truncation():
    g, p, X, W2, Y = _vh_case("v01_vh_svd")
    V, H, VHr = cp_oracle.vh_decompose_oracle(W2, rank=p["rank"])
    assert np.array_equal(V, g["V"]) and np.array_equal(H, g["H"]) and np.array_equal(VHr, g["VHr"])


def test_vh_oracle_reproduces_reference_refit():
    g, p, X, W2, Y = _vh_case("v02_vh_refit")
    V, H, VHr, b = cp_oracle.vh_decompose_oracle(W2, rank=p["rank"], X=X, Y=Y)
    assert np.array_equal(V, g["V"])
    assert relfro(H, g["H"]) <= 1e-12 and relfro(VHr, g["VHr"]) <= 1e-12 and relfro(b, g["b"]) <= 1e-12


def test_itq_oracle_reproduces_reference():
    g = np.load(os.path.join(GOLDEN_DIR, "i01_itq.npz"))
    p = json.loads(str(g["params"]))
    X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"])
    feature = Y + p["noise"] * np.random.RandomState(p["layer_id"]).randn(*Y.shape)
    W1, Wo2, B, W12 = cp_oracle.itq_decompose_oracle(feature, Y, W2.astype(np.float64), p["rank"],
                                                     bias=B2.astype(np.float64))
    for a, name in ((W1, "W1"), (Wo2, "W2"), (B, "B"), (W12, "W12")):
        assert a.shape == g[name].shape and relfro(a, g[name]) <= 1e-10, name


def test_soft_threshold_ties_separate_every_rounding_variant_and_sklearns_own_two_paths():
    """tests/golden/t01_ties.npz (oracle/gen_golden_ties.py): a coordinate exactly on the edge of its dead zone.  The four
    rounding variants of the Gram-form update decide its support differently, scikit-learn's Gram form differs from its data
    form -- which is what the reference runs (Lasso(...).fit(Z, reY), precompute=False) -- and the data form agrees with
    flags 0 on some ties and with flags 3 on others: no Gram-form variant can promise the reference's mask AT a tie.  The
    C restatement reproduces the stored coefficients bit for bit; scikit-learn's data form (when importable) reproduces the
    stored support."""
    g = np.load(os.path.join(GOLDEN_DIR, "t01_ties.npz"))
    seed = int(g["seed"])
    agree = np.zeros(4, dtype=int)
    for t in range(g["l1"].shape[0]):
        Z, y, l1 = g["Z"][t], g["y"][t], float(g["l1"][t])
        Q, q, yy = Z.T @ Z, Z.T @ y, float(y @ y)
        sups = []
        for k, (recip, delta) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
            w = np.zeros(2)
            w, _, it = cp_oracle.enet_cd_gram(w, l1, 0.0, Q, q, yy, seed=seed, recip=recip, delta=delta)
            assert np.array_equal(w, g["w"][t, k]) and it == int(g["n_iter"][t, k])
            sups.append(tuple(w != 0))
            agree[k] += sups[-1] == tuple(g["sk_data"][t] != 0)
        assert len(set(sups)) > 1                                   # the variants disagree on every stored tie
    assert 0 < agree[0] < g["l1"].shape[0] and 0 < agree[3] < g["l1"].shape[0]   # neither flags 0 nor flags 3 always sides with the data form
    assert any(tuple(g["sk_data"][t] != 0) != tuple(g["sk_gram"][t] != 0) for t in range(g["l1"].shape[0]))
    try:
        from sklearn.linear_model import Lasso
    except ImportError:
        return
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(g["l1"].shape[0]):
            m = Lasso(alpha=float(g["l1"][t]) / g["Z"].shape[1], fit_intercept=False, selection="random", precompute=False,
                      random_state=np.random.RandomState(0), tol=1e-4, max_iter=1000).fit(g["Z"][t], g["y"][t])
            assert tuple(m.coef_ != 0) == tuple(g["sk_data"][t] != 0)


def test_three_operation_division_is_correctly_rounded():
    """tests/host/test_markstein.c: q1 = fma(fma(-b, a r, a), r, a r) with r = RN(1 / b) equals a / b on 1e8 random,
    adversarial and structured operand pairs (the chain wave of csrc/cd_team.hip divides this way)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = "/tmp/cp_test_markstein"
    subprocess.check_call(["gcc", "-O2", "-mfma", os.path.join(ROOT, "tests", "host", "test_markstein.c"), "-lm", "-o", exe])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "random: 0 mismatches" in out.stdout and "adversarial: 0 mismatches" in out.stdout \
        and "structured: 0 mismatches" in out.stdout, out.stdout


def test_oracle_follows_the_reference_alpha_chain_on_the_first_layers_of_the_vgg16_job():
    """C01_vgg16_alpha_chain.npz = the unmodified reference over V01..V12 with cfgs.alpha carried (oracle/gen_golden.py
    --chain).  The port (C Gram-form CD + numpy lstsq) run the same way reproduces the chain on its first four layers
    (c = 64, 64, 128, 128: seconds on CPU; the whole chain is the GPU test's): masks, per-fit logs, carried alphas, RNG."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import jobs
    g = np.load(os.path.join(GOLDEN_DIR, "C01_vgg16_alpha_chain.npz"))
    specs = jobs.vgg16_4x()
    assert json.loads(str(g["names"])) == [s["name"] for s in specs]
    assert float(g["alpha_in"][0]) == 1e-3 and np.array_equal(g["alpha_in"][1:], g["alpha_out"][:-1])
    assert len(set(g["alpha_out"].tolist())) >= 4         # the carry is live: the layers do end at different alphas
    alpha = 1e-3
    for i, spec in enumerate(specs[:4]):
        X, W2, Y, B2 = jobs.synth(spec)
        np.random.seed(1234 + spec["layer_id"])
        log = []
        idxs, newW2, newB2, alpha = cp_oracle.dictionary_oracle(X.astype(np.float64), W2, Y, spec["rank"], B2, alpha_in=alpha,
                                                                lasso="c_gram", ls="numpy", log=log)
        rng_next = int(np.random.randint(0, 2147483647))
        fits = np.array([(f[1], f[2], f[3]) for f in log if f[0] == "fit"], dtype=np.float64).reshape(-1, 3)
        assert np.array_equal(idxs, g["idxs_%02d" % i])
        assert np.array_equal(fits, g["fits_%02d" % i])
        assert alpha == float(g["alpha_out"][i]) and rng_next == int(g["rng_next"][i])
        wm = newW2.reshape(newW2.shape[0], -1)
        assert relfro(wm @ cp_oracle.sketch_matrix(wm.shape[1]), g["newW2_sketch_%02d" % i]) <= 1e-8
