"""A/B of process-wide experiment switches (cp_debug_knob) on the resident whole-network job, in ONE process, alternating.

    python tools/probes/job_knobs.py [--job vgg16] [--reps 3] [--jobs 20] 0=0 0=9 0=12,1=8 ...

Every argument is one setting: comma-separated knob=value pairs (knobs not named are 0).  Prints the job time of every
setting per repetition and the best / median; masks of every setting are compared with the first."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchkit import common as _common   # noqa: E402,F401  (GPU_MAX_HW_QUEUES, sys.path)
from benchkit.common import CD_FLAGS, cpjobs   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--job", default="vgg16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--jobs", type=int, default=20)
    ap.add_argument("--pre", type=int, default=None, help="ResidentLayerSet(precompute_heaviest=...)")
    ap.add_argument("--per-stream", default="", help="layers per stream by width, e.g. 64:2,128:2 (default 1)")
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    import ctypes
    from cpmi355 import shard
    specs = cpjobs.JOBS[a.job]()
    per_stream = 1 if a.job != "resnet50" else {"default": 2, 2048: 1}
    if a.per_stream:
        per_stream = {"default": 1} if not isinstance(per_stream, dict) else dict(per_stream)
        for item in a.per_stream.split(","):
            k, v = item.split(":")
            per_stream[int(k)] = int(v)
    rset = shard.ResidentLayerSet(0, specs, lambda sp: cpjobs.synth(sp)[:3], per_stream=per_stream, flags=CD_FLAGS, borrow_results=True,
                                  precompute_heaviest=a.pre)
    roots = [ch["ctxs"][0] for ch in rset.chunks]
    lib = roots[0].lib
    lib.cp_debug_knob.argtypes, lib.cp_debug_knob.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int

    def apply(setting):
        for k in range(8):
            lib.cp_debug_knob(k, 0)
        rset.narrow_delay_ms = 0.0
        for item in setting.split(","):
            k, v = item.split("=")
            if k == "d":                     # host-side: the narrow layers' threads start that many ms after start()
                rset.narrow_delay_ms = float(v)
            else:
                lib.cp_debug_knob(int(k), int(v))

    def sync():
        for cx in roots:
            cx.sync()

    for _ in range(3):
        rset()
    sync()
    ref = None
    times = {s: [] for s in a.settings}
    for rep in range(a.reps):
        for s in a.settings:
            apply(s)
            for _ in range(3):
                rset()
            sync()
            t0 = time.perf_counter()
            for _ in range(a.jobs):
                res = rset()
            sync()
            times[s].append((time.perf_counter() - t0) / a.jobs * 1e3)
            masks = [np.array(r[0]) for r in res]
            if ref is None:
                ref = masks
            elif not all(np.array_equal(x, y) for x, y in zip(ref, masks)):
                print("MASKS DIFFER at setting", s)
    for s in a.settings:
        print("%-24s best %.3f  median %.3f  runs %s" % (s, min(times[s]), float(np.median(times[s])), " ".join("%.3f" % t for t in times[s])))
    rset.close()


if __name__ == "__main__":
    main()
