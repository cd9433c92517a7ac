"""Run every golden case through the drop-in dictionary() with a given CD flag set; report mask/fit-log parity."""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import cp_oracle, cpmi355
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = cpmi355.Context(0)
bad = 0
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
    g = np.load(path); p = json.loads(str(g["params"]))
    if p["rank"] == p["c"]: continue
    X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"], dead=p.get("dead", 0), residual=p.get("residual", False))
    prob = cpmi355.LayerProblem(ctx, X, W2, Y, flags=flags)
    rng = np.random.RandomState(1234 + p["layer_id"])
    idxs, W, b, a = cpmi355.prune_layer(prob, p["rank"], p.get("alpha_in", 1e-3), rank_tol=p.get("rank_tol", .1), rng=rng, ridge=p.get("fc_ridge", 0.0))
    fits = np.array(prob.fits, dtype=np.float64).reshape(-1, 3)
    ok = np.array_equal(idxs, g["idxs"]) and fits.shape == g["fits"].shape and np.array_equal(fits, g["fits"])
    bad += not ok
    print(os.path.basename(path), "OK" if ok else "MISMATCH", len(fits))
    prob.free()
print("flags", flags, "mismatches", bad)
