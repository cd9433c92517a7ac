"""benchkit.extras -- the other configurations of BASELINE.json inside the default `bench.py --gpus 1` run, each as ONE short
leg after the headline job: the resnet50 job (configs[3]), the vgg16_5x job (configs[4]) and one R3 pass (configs[2]'s
"(3C) prune": VH -> ITQ -> prune per conv).  Same machinery as the headline (cpmi355.shard.ResidentLayerSet, every layer
checked against its reference golden), a fraction of a second of timed jobs each; the full-length lines of these workloads
remain `bench.py --workload resnet50 | vgg16_5x | r3`."""
import time

import numpy as np

from .common import CD_FLAGS, cpjobs, golden_check


def per_stream_for(job, requested=0):
    """layers per stream / cp_prune_layers call.  resnet50: the two 2048-channel selections own the critical path (their alpha
    searches, ~25 ms each): a stream each, so that neither waits for the other's Gram and refit (34.9 against 37.5 ms per
    job); the other widths two layers per stream"""
    if requested:
        return requested
    return {"default": 2, 2048: 1} if job == "resnet50" else 1


def short_job(device, job, min_seconds=1.0, min_jobs=4, warmup_seconds=0.6):
    """-> {layers_per_s, job_ms, jobs_timed, layers, mask_parity, weights_rel_frobenius_max}: one instance of `job`, operands
    resident, untimed jobs for >= warmup_seconds (the leg follows the CPU baseline: the GPU sat idle for ~45 s, its clocks are
    down, and the contexts' workspaces still grow during the first jobs), then back-to-back jobs for >= min_seconds; every
    layer against its reference golden."""
    from cpmi355 import shard
    specs = cpjobs.JOBS[job]()
    rset = shard.ResidentLayerSet(device, specs, lambda s: cpjobs.synth(s)[:3], per_stream=per_stream_for(job), flags=CD_FLAGS,
                                  borrow_results=True)
    try:
        roots = [ch["ctxs"][0] for ch in rset.chunks]
        t_w, per = time.perf_counter(), 1.0
        while True:
            t0 = time.perf_counter()
            rset()
            for cx in roots:
                cx.sync()
            per = time.perf_counter() - t0
            if time.perf_counter() - t_w >= warmup_seconds:
                break
        jobs = max(min_jobs, int(np.ceil(min_seconds / max(per, 1e-4))))
        t0 = time.perf_counter()
        for _ in range(jobs):
            res = rset()
        for cx in roots:
            cx.sync()
        elapsed = time.perf_counter() - t0
        parity, werr_max, checked = True, 0.0, 0
        for spec, (idxs, W, _) in zip(specs, res):
            same, werr = golden_check(spec["name"], idxs, np.asarray(W))
            if same is None:
                continue
            checked += 1
            parity = parity and bool(same) and werr is not None and werr <= 1e-5
            if werr is not None:
                werr_max = max(werr_max, werr)
        return {"layers_per_s": round(len(specs) * jobs / elapsed, 1), "job_ms": round(elapsed / jobs * 1e3, 2), "jobs_timed": jobs,
                "layers": len(specs), "mask_parity": bool(parity) if checked else None, "layers_with_golden": checked,
                "weights_rel_frobenius_max": float("%.2e" % werr_max)}
    finally:
        rset.close()
