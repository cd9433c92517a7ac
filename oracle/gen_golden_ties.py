"""Constructed soft-threshold ties (tests/golden/t01_ties.npz).

A LASSO coordinate sits EXACTLY on the boundary of its dead zone at the fixed point of the iteration: two features, integer
data, and an l1 weight placed (to the last bit) where |q1 - Q01 w0| = l1 with w1 = 0.  There every rounding-level variant
of the coordinate update decides the support of w differently, and so do scikit-learn's own two code paths:

    data form   sklearn _cd_fast.enet_coordinate_descent        <- what the reference runs (Lasso(...).fit(Z, reY),
                                                                   precompute=False, lib/decompose.py:449, 456)
    Gram form   sklearn _cd_fast.enet_coordinate_descent_gram   <- Lasso(precompute=True)
    oracle / device Gram form, flags 0..3 (CP_CD_RECIPROCAL | CP_CD_DELTA)

So no Gram-form implementation -- flags 0 included -- can promise the reference's mask AT a tie; what can be promised, and
is tested, is (a) identical masks / epoch counts on every non-degenerate input (all dictionary() goldens) and (b) that the
device follows its CPU restatement bit for bit even here (tests/test_gpu_parity.py).  This script searches the ties, runs
scikit-learn both ways, and stores everything; python oracle/gen_golden_ties.py."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cp_oracle  # noqa: E402

VARIANTS = ((0, 0), (1, 0), (0, 1), (1, 1))     # (recip, delta) = flags 0, 1, 2, 3


def supports(Q, q, yy, l1, seed):
    out = []
    for recip, delta in VARIANTS:
        w = np.zeros(Q.shape[0])
        w, _, it = cp_oracle.enet_cd_gram(w, l1, 0.0, Q, q, yy, seed=seed, recip=recip, delta=delta)
        out.append((w.copy(), it))
    return out


def main(want=6):
    from sklearn.linear_model import Lasso
    warnings.filterwarnings("ignore")
    rs = np.random.RandomState(1)
    seed = int(np.random.RandomState(0).randint(0, 2147483647))      # what Lasso draws from random_state=RandomState(0)
    cases = []
    for trial in range(2000):
        M = 40
        Z = rs.randint(-6, 7, size=(M, 2)).astype(np.float64)
        Z -= Z.mean(0)
        y = rs.randint(-9, 10, size=M).astype(np.float64)
        y -= y.mean()
        Q, q, yy = Z.T @ Z, Z.T @ y, float(y @ y)
        a, b = Q[0, 0], Q[0, 1]
        if abs(b) >= 0.9 * min(a, Q[1, 1]) or b == 0:
            continue
        s0, hit = np.sign(q[0]), None
        for s1 in (1.0, -1.0):
            l1 = (s1 * q[1] - s1 * s0 * b * abs(q[0]) / a) / (1 - s1 * s0 * b / a)
            if not 0 < l1 < abs(q[0]):
                continue
            for k in range(-40, 41):
                lk = l1
                for _ in range(abs(k)):
                    lk = np.nextafter(lk, np.inf if k > 0 else -np.inf)
                sup = supports(Q, q, yy, lk, seed)
                if len({tuple(w != 0) for w, _ in sup}) > 1:
                    hit = (lk, sup)
                    break
            if hit:
                break
        if not hit:
            continue
        lk, sup = hit
        sk = {}
        for pre in (False, True):
            m = Lasso(alpha=lk / M, fit_intercept=False, selection="random", precompute=pre,
                      random_state=np.random.RandomState(0), tol=1e-4, max_iter=1000)
            m.fit(Z, y)
            sk[pre] = (m.coef_.copy(), int(m.n_iter_))
        cases.append(dict(Z=Z, y=y, l1=lk, w=np.stack([w for w, _ in sup]), n_iter=np.array([it for _, it in sup]),
                          sk_data=sk[False][0], sk_data_iter=sk[False][1], sk_gram=sk[True][0], sk_gram_iter=sk[True][1]))
        print("tie %d: l1 = %r  oracle supports %s  sklearn data %s gram %s" % (
            len(cases), float(lk), [tuple(bool(v) for v in w != 0) for w, _ in sup],
            tuple(bool(v) for v in sk[False][0] != 0), tuple(bool(v) for v in sk[True][0] != 0)))
        if len(cases) >= want:
            break
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "t01_ties.npz")
    np.savez(out, seed=seed, Z=np.stack([c["Z"] for c in cases]), y=np.stack([c["y"] for c in cases]),
             l1=np.array([c["l1"] for c in cases]), w=np.stack([c["w"] for c in cases]),
             n_iter=np.stack([c["n_iter"] for c in cases]), sk_data=np.stack([c["sk_data"] for c in cases]),
             sk_gram=np.stack([c["sk_gram"] for c in cases]),
             sk_data_iter=np.array([c["sk_data_iter"] for c in cases]), sk_gram_iter=np.array([c["sk_gram_iter"] for c in cases]))
    print("wrote", out)


if __name__ == "__main__":
    main()
