"""TEST INFRASTRUCTURE -- golden vectors for VH_decompose / nonlinear_fc from the REAL reference
(/root/reference/lib/decompose.py:85-146, 671-685), run here in the build container.
    python oracle/gen_golden_vh.py      ->  tests/golden/v01_vh_svd.npz, v02_vh_refit.npz
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cp_oracle  # noqa: E402
import ref_loader  # noqa: E402
from gen_golden import versions  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")
ITQ_CASES = {
    "i01_itq": dict(layer_id=18, N=1500, c=24, n=40, k=3, rank=20, noise=0.05),
    # conv3 size (256 x 256 x 3 x 3, N = 5000, rank as the reference's R3 uses for conv3_1: int(83 * 4 / 3) = 110)
    "i02_itq_conv3": dict(layer_id=48, N=5000, c=256, n=256, k=3, rank=110, noise=0.05, large=True),
}
CASES = {
    "v01_vh_svd": dict(layer_id=16, N=64, c=32, n=48, k=3, rank=40, with_x=False),
    "v02_vh_refit": dict(layer_id=17, N=1500, c=24, n=40, k=3, rank=30, with_x=True),
    # conv3 size with the ReLU-aware refit of H (rank 110 = the reference's conv3_1 entry, net.py:1313, 1323-1326)
    "v03_vh_refit_conv3": dict(layer_id=47, N=5000, c=256, n=256, k=3, rank=110, with_x=True, large=True),
}


def main():
    D, _ = ref_loader.load()
    only = sys.argv[1:]
    for name, p in CASES.items():
        if only and name not in only:
            continue
        X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"])
        t0 = time.perf_counter()
        if p["with_x"]:
            V, H, VHr, b = D.VH_decompose(W2.astype(np.float64), rank=p["rank"], X=X.astype(np.float64), Y=Y)
        else:
            V, H, VHr = D.VH_decompose(W2.astype(np.float64), rank=p["rank"])
            b = np.zeros(0)
        dt = time.perf_counter() - t0
        f = (lambda a: a.astype(np.float32)) if p.get("large") else (lambda a: a)   # 6e-8 << the 1e-5 budget
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), params=json.dumps(p), versions=json.dumps(versions()), V=f(V), H=f(H),
                            VHr=f(VHr), b=b, ref_seconds=dt)
        print("%-14s V%s H%s VHr%s  %.2fs" % (name, V.shape, H.shape, VHr.shape, dt))
    for name, p in ITQ_CASES.items():
        if only and name not in only:
            continue
        X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"])
        feature = Y + p["noise"] * np.random.RandomState(p["layer_id"]).randn(*Y.shape)   # the approximated layer's output
        t0 = time.perf_counter()
        W1, Wo2, B, W12 = D.ITQ_decompose(feature, Y, W2.astype(np.float64), p["rank"], bias=B2.astype(np.float64))
        dt = time.perf_counter() - t0
        f = (lambda a: a.astype(np.float32)) if p.get("large") else (lambda a: a)
        np.savez_compressed(os.path.join(GOLDEN_DIR, name + ".npz"), params=json.dumps(p), versions=json.dumps(versions()),
                            W1=f(W1), W2=f(Wo2), B=B, W12=f(W12), ref_seconds=dt)
        print("%-14s W1%s W2%s B%s W12%s  %.2fs" % (name, W1.shape, Wo2.shape, B.shape, W12.shape, dt))


if __name__ == "__main__":
    main()
