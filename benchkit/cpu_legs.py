"""benchkit.cpu_legs -- the `cpu_baseline` legs of bench.py: the CPU port of the reference path (oracle/cp_oracle.py driving
scikit-learn's own Lasso / LinearRegression: the arithmetic the reference runs) timed on this box's host cores.  The ONLY
module of the bench that imports anything under oracle/ -- as the thing timed next to the product, never inside it."""
import os
import sys
import time

import numpy as np

from .common import ROOT, cpjobs, cpu_model

def cpu_port_seconds(specs, threads=None, carry_alpha=False, masks_out=None):
    """CPU port of the reference path: seconds per layer.  specs: cpmi355.jobs spec dicts.  threads: BLAS / OpenMP
    thread limit (threadpoolctl) or None for the library default (all cores).  carry_alpha: every layer starts its search
    from the alpha the previous one ended with (cfgs.alpha, /root/reference/lib/decompose.py:491, 626-627) instead of 1e-3.
    masks_out: list that receives (idxs, alpha_out) per layer."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from threadpoolctl import threadpool_limits
    secs, alpha = [], 1e-3
    for spec in specs:
        X, W2, Y, B2 = cpjobs.synth(spec)
        X64 = X.astype(np.float64)
        np.random.seed(1234 + spec["layer_id"])
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            out = cp_oracle.dictionary_oracle(X64, W2, Y, spec["rank"], B2, alpha_in=alpha if carry_alpha else 1e-3,
                                              lasso="sklearn", ls="sklearn")
            secs.append(time.perf_counter() - t0)
        if carry_alpha:
            alpha = out[3]
        if masks_out is not None:
            masks_out.append((out[0], out[3]))
    return secs


def cpu_best_threads(specs):
    """-> (thread count that minimises the port's time on `specs`, {threads: seconds})"""
    ncpu = os.cpu_count() or 1
    sweep = {}
    for t in sorted({1, min(8, ncpu), min(32, ncpu), ncpu}):
        sweep[t] = round(sum(cpu_port_seconds(specs, threads=t)), 3)
    return min(sweep, key=lambda t: sweep[t]), sweep


def cpu_baseline_object(specs, sample, per_layer, job_ms, full, carry_alpha=False, masks_out=None):
    """cpu_baseline of the JSON line: the port on `sample` at the best BLAS thread count of this box."""
    best, sweep = cpu_best_threads(sample[:3])
    secs = cpu_port_seconds(sample, threads=best, carry_alpha=carry_alpha, masks_out=masks_out)
    gpu_ms_same = sum(per_layer[s["name"]]["ms_alone"] for s in sample if s["name"] in per_layer)
    out = {"value": round(len(sample) / sum(secs), 4), "unit": "layers/s", "cores": int(best), "kind": "port",
           "sample": "%s of the job's %d layers, one pass of the CPU port: sklearn Lasso (single-threaded CD) + LinearRegression/"
                     "gelsd at %d BLAS threads (the fastest of the sweep), %.1f s" % (
                         "all" if full else "the %d cheapest" % len(sample), len(specs), best, sum(secs)),
           "sample_layers": [s["name"][:3] for s in sample], "per_layer_s": [round(x, 2) for x in secs],
           "blas_thread_sweep_s": {"layers": [s["name"][:3] for s in sample[:3]], "seconds_by_threads": sweep},
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    if gpu_ms_same > 0:
        out["gpu_ms_same_layers_one_at_a_time"] = round(gpu_ms_same, 2)
        out["speedup_same_layers_latency"] = round(sum(secs) * 1e3 / gpu_ms_same, 1)
    if full:
        out["job_seconds_cpu"] = round(sum(secs), 2)
        out["job_speedup_wall_clock"] = round(sum(secs) * 1e3 / job_ms, 1)
    return out



def r3_cpu_baseline(plan, data, per, sample=(0, 1)):
    """cpu_baseline of `--workload r3`: the scipy gesvd / sklearn restatement of the three steps (oracle/cp_oracle.py) on the
    first convs of the pass (conv1_2, conv2_1: about 20-40 s of CPU work at 8 BLAS threads), next to the GPU time of the same"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from threadpoolctl import threadpool_limits
    sample = list(sample)
    secs = {}
    with threadpool_limits(limits=8):
        for i in sample:
            st, d = plan[i], data[i]
            np.random.seed(2000 + i)
            t0 = time.perf_counter()
            V, H, VHr, b = cp_oracle.vh_decompose_oracle(d["W"].astype(np.float64), rank=st["rank"], X=d["X"].astype(np.float64), Y=d["Y"])
            t1 = time.perf_counter()
            cp_oracle.itq_decompose_oracle(d["feat"], d["Y"], H, st["rank"], bias=b, Wr=VHr)
            t2 = time.perf_counter()
            if st["prune"]:
                cp_oracle.dictionary_oracle(d["Xo"].astype(np.float64), d["W2"], d["Y2"], st["d_c"], alpha_in=1e-3,
                                            lasso="sklearn", ls="sklearn")
            t3 = time.perf_counter()
            secs[st["name"]] = dict(vh_s=round(t1 - t0, 2), itq_s=round(t2 - t1, 2), prune_s=round(t3 - t2, 2))
    cpu_s = sum(sum(v.values()) for v in secs.values())
    gpu_ms = sum(per[plan[i]["name"]]["vh_ms"] + per[plan[i]["name"]]["itq_ms"] + (per[plan[i]["name"]]["prune_ms"] or 0.0) for i in sample)
    return {"value": round(len(sample) / cpu_s, 4), "unit": "layers/s", "cores": 8, "kind": "port",
                           "sample": "the first %d convs (%s): scipy gesvd / sklearn restatement of the three steps (oracle/cp_oracle.py), "
                                     "8 BLAS threads: %.1f s" % (len(sample), ", ".join(plan[i]["name"] for i in sample), cpu_s),
                           "per_conv_s": secs, "gpu_ms_same_convs": round(gpu_ms, 2),
                           "speedup_same_convs": round(cpu_s * 1e3 / gpu_ms, 1), "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
