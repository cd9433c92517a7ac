"""TEST INFRASTRUCTURE ONLY -- pins the oracle restatement.

For every golden case (outputs of the unmodified reference, tests/golden/*.npz) run
``cp_oracle.dictionary_oracle`` with each engine and report agreement:
  engine sklearn : must be bit-identical (same third-party code path as the reference)
  engine c_data  : C restatement of the data-form CD   -> same per-fit (nnz, n_iter), mask
  engine c_gram  : C restatement of the Gram-form CD   -> same per-fit (nnz, n_iter), mask
  ls numpy       : truncated-SVD min-norm restatement   -> rel. Frobenius <= 1e-9
Usage: python oracle/validate_oracle.py [--large]
"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cp_oracle  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")


def relfro(a, b):
    d = np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    nb = np.linalg.norm(np.asarray(b, dtype=np.float64))
    return d / nb if nb > 0 else d


def run_case(path, lasso, ls):
    g = np.load(path)
    p = json.loads(str(g["params"]))
    X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"],
                                         dead=p.get("dead", 0), residual=p.get("residual", False))
    np.random.seed(1234 + p["layer_id"])
    log = []
    idxs, newW2, newB2, alpha_out = cp_oracle.dictionary_oracle(
        X.astype(np.float64), W2, Y, p["rank"], B2, alpha_in=p.get("alpha_in", 1e-3),
        rank_tol=p.get("rank_tol", .1), lasso=lasso, ls=ls, ridge=p.get("fc_ridge", 0.0), log=log,
        refit="nonlinear" if p.get("nonlinear_fc") else ("none" if p.get("nofc") else "linear"))
    rng_next = int(np.random.randint(0, 2147483647))
    fits = np.array([(f[1], f[2], f[3]) for f in log if f[0] == "fit"], dtype=np.float64).reshape(-1, 3)
    samples = [f[1] for f in log if f[0] == "samples"][0]
    ok_mask = bool(np.array_equal(idxs, g["idxs"]))
    ok_fits = fits.shape == g["fits"].shape and bool(np.array_equal(fits, g["fits"]))
    ok_rng = rng_next == int(g["rng_next"])
    ok_samp = bool(np.array_equal(samples, g["samples"]))
    ew = relfro(newW2, g["newW2"]) if ok_mask else float("nan")
    eb = relfro(newB2, g["newB2"]) if ok_mask else float("nan")
    return dict(mask=ok_mask, fits=ok_fits, rng=ok_rng, samples=ok_samp, eW=ew, eB=eb,
                alpha=(alpha_out == float(g["alpha_out"])))


def main():
    large = "--large" in sys.argv
    bad = 0
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(path)[:-4]
        if name.startswith("L") and not large:
            continue
        for lasso, ls in (("sklearn", "sklearn"), ("c_data", "numpy"), ("c_gram", "numpy")):
            r = run_case(path, lasso, ls)
            good = r["mask"] and r["fits"] and r["rng"] and r["samples"] and r["alpha"] and \
                r["eW"] <= 1e-5 and r["eB"] <= 1e-5
            bad += not good
            print("%-24s %-8s/%-7s mask %d fits %d rng %d alpha %d  eW %.2e eB %.2e %s" % (
                name, lasso, ls, r["mask"], r["fits"], r["rng"], r["alpha"], r["eW"], r["eB"],
                "" if good else "  <-- MISMATCH"))
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
