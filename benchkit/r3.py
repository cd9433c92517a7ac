"""benchkit.r3 -- `--workload r3`: the three steps Net.R3 runs per conv of VGG-16 (VH -> ITQ -> prune)."""
import time

import numpy as np

from .common import N_SAMPLES

# ==================================================================================================================
# workload: r3 -- the three steps Net.R3 runs per conv of VGG-16
# ==================================================================================================================
VGG16_CONVS = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128), ("conv3_1", 128, 256),
               ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512),
               ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]
# /root/reference/lib/net.py:1309-1327: the 3C-4x ranks (conv5_x as listed, the others x 4 / dic.keep with dic.keep = 3)
R3_RANK = {"conv1_2": 17, "conv2_1": 37, "conv2_2": 47, "conv3_1": 83, "conv3_2": 89, "conv3_3": 106, "conv4_1": 175,
           "conv4_2": 192, "conv4_3": 227, "conv5_1": 398, "conv5_2": 390, "conv5_3": 379}
R3_PRUNED = ("conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2")   # alldic + pooldic (net.py:1307-1308)


def r3_plan():
    """Per conv of the reference's R3 loop (net.py:1339-1459, conv1_2 .. conv5_3): the shapes VH_decompose, ITQ_decompose and
    dictionary_kernel see -- input channels already reduced when the conv was the consumer of an earlier pruning."""
    plan, kept_in = [], {}
    for i, (name, c, n) in enumerate(VGG16_CONVS[1:], start=1):
        rank = R3_RANK[name] if name.startswith("conv5") else int(R3_RANK[name] * 4.0 / 3.0)
        d_c = max(int(n / 1.15), rank)
        step = dict(name=name, c=kept_in.get(name, c), n=n, rank=rank, d_c=d_c, prune=None)
        if name in R3_PRUNED and i + 1 < len(VGG16_CONVS):
            nxt, _, n_next = VGG16_CONVS[i + 1]
            step["prune"] = dict(consumer=nxt, n_next=n_next)
            kept_in[nxt] = d_c
        plan.append(step)
    return plan


def r3_operands(step, seed, N=N_SAMPLES):
    rs = np.random.RandomState(seed)
    c, n = step["c"], step["n"]
    X = np.maximum(rs.randn(N, c, 3, 3), 0).astype(np.float32)
    W = (rs.randn(n, c, 3, 3) * 0.05).astype(np.float32)
    Y = X.reshape(N, -1).astype(np.float64) @ W.reshape(n, -1).T.astype(np.float64) + 0.01 * rs.randn(N, n)
    feat = Y + 0.02 * rs.randn(N, n)          # the response of the spatially decomposed conv at the same points
    out = dict(X=X, W=W, Y=Y, feat=feat)
    if step["prune"]:
        n2 = step["prune"]["n_next"]
        Xo = np.maximum(rs.randn(N, n, 3, 3), 0).astype(np.float32)
        W2 = (rs.randn(n2, n, 3, 3) * 0.05).astype(np.float32)
        out.update(Xo=Xo, W2=W2, Y2=Xo.reshape(N, -1).astype(np.float64) @ W2.reshape(n2, -1).T.astype(np.float64) + 0.01 * rs.randn(N, n2))
    return out


def r3_setup(device):
    """-> (context, plan, per-conv operands, one_pass(record=None)): one_pass runs VH -> ITQ -> (prune) for the 12 convs through
    the drop-in functions of lib/decompose.py, from host arrays, and appends per-conv milliseconds to `record`"""
    import cpmi355
    import lib.cfgs as cfgs
    import lib.decompose as D
    ctx = cpmi355.default_context(device)
    plan = r3_plan()
    data = [r3_operands(st, 4000 + i) for i, st in enumerate(plan)]

    def one_pass(record=None):
        cfgs.alpha = 1e-3
        for i, (st, d) in enumerate(zip(plan, data)):
            np.random.seed(2000 + i)
            t0 = time.perf_counter()
            V, H, VHr, b = D.VH_decompose(d["W"], rank=st["rank"], DEBUG=True, X=d["X"], Y=d["Y"])
            ctx.sync()
            t1 = time.perf_counter()
            D.ITQ_decompose(d["feat"], d["Y"], H, st["rank"], bias=b, DEBUG=0, Wr=VHr)
            ctx.sync()
            t2 = time.perf_counter()
            if st["prune"]:
                D.dictionary(d["Xo"].astype(np.float64, copy=False), d["W2"], d["Y2"], rank=st["d_c"])
                ctx.sync()
            t3 = time.perf_counter()
            if record is not None:
                r = record.setdefault(st["name"], dict(vh=[], itq=[], prune=[]))
                r["vh"].append((t1 - t0) * 1e3)
                r["itq"].append((t2 - t1) * 1e3)
                r["prune"].append((t3 - t2) * 1e3)

    return ctx, plan, data, one_pass


def r3_short_pass(device, passes=2):
    """the R3 leg of the default line: one untimed pass, then `passes` timed ones -> seconds per pass and per step class (per
    conv the fastest of the passes, as bench_r3 reports them)"""
    ctx, plan, data, one_pass = r3_setup(device)
    one_pass()
    rec = {}
    for _ in range(passes):
        one_pass(rec)
    tot = lambda key: round(sum(min(v[key]) for v in rec.values()) / 1e3, 3)   # noqa: E731
    vh, itq, prune = tot("vh"), tot("itq"), tot("prune")
    return {"pass_s": round(vh + itq + prune, 3), "vh_s": vh, "itq_s": itq, "prune_s": prune, "convs": len(plan),
            "passes_timed": passes,
            "note": "Net.R3's three steps per conv of VGG-16 (VH -> ITQ -> dictionary for the 7 pruned convs), 3C-4x ranks, "
                    "N = 5000, host arrays in and out; bench.py --workload r3 is the full line"}


def bench_r3(args, env):
    """`--workload r3`: what Net.R3 (/root/reference/lib/net.py:1292-1471) runs per conv of VGG-16 -- spatial decomposition
    (VH_decompose with the ReLU-aware refit of H: 50 alternations), channel decomposition (ITQ_decompose: 50 alternations, each
    a rank-truncated SVD) and, for the 7 convs of alldic / pooldic, channel pruning against the next conv (dictionary) -- at
    the reference's 3C-4x ranks, N = 5000 sampled points per conv, through the drop-in functions of lib/decompose.py from
    host arrays.  Synthetic per-conv operands; the forward passes that re-extract features between the steps (Caffe in the
    reference, a torch provider in lib/provider.py) are not part of the timed work."""
    ctx, plan, data, one_pass = r3_setup(env.local_rank)

    for _ in range(max(1, min(args.warmup, 2))):
        one_pass()
    rec = {}
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(rec)
    elapsed = time.perf_counter() - t0
    job_ms = elapsed / steps * 1e3
    per = {k: {"c": st["c"], "n": st["n"], "rank": st["rank"], "d_c": st["d_c"] if st["prune"] else None,
               "vh_ms": round(min(rec[k]["vh"]), 2), "itq_ms": round(min(rec[k]["itq"]), 2),
               "prune_ms": round(min(rec[k]["prune"]), 2) if st["prune"] else None}
           for k, st in ((st["name"], st) for st in plan)}
    out = {"metric": "conv layers decomposed + pruned/sec (VGG-16 3C 4x steps of Net.R3, 5k samples)",
           "value": round(len(plan) * steps / elapsed, 3), "unit": "layers/s", "n_gpus": 1, "steps": steps,
           "warmup": max(1, min(args.warmup, 2)), "ms_per_step": round(job_ms, 2), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "r3: the 12 convs conv1_2 .. conv5_3 of VGG-16, per conv VH_decompose (rank-truncated SVD + 50 ReLU-aware "
                                  "refits) -> ITQ_decompose (50 alternations) -> dictionary() for the 7 convs the reference prunes (alldic + "
                                  "pooldic), 3C-4x ranks of /root/reference/lib/net.py:1309-1327, N = 5000; one after another as R3 does; "
                                  "1 step = 1 pass over the 12 convs",
                      "layers_per_job": len(plan)},
           "job_ms": round(job_ms, 2),
           "stage_ms_per_job": {"spatial_decomposition (VH)": round(sum(v["vh_ms"] for v in per.values()), 2),
                                "channel_decomposition (ITQ)": round(sum(v["itq_ms"] for v in per.values()), 2),
                                "channel_pruning (dictionary)": round(sum(v["prune_ms"] or 0.0 for v in per.values()), 2)},
           "per_conv": per, "roofline": None,
           "note": "host-inclusive: every call starts from NumPy arrays, as Net.R3 hands them over; latency-bound by the Jacobi "
                   "sweeps of the SVDs (svd_jacobi.hip) -- no roofline kernel is named for this workload"}
    if not args.no_cpu_baseline:
        from .cpu_legs import r3_cpu_baseline
        out["cpu_baseline"] = r3_cpu_baseline(plan, data, per)
    return out

