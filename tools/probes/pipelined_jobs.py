"""Several independent instances of the whole-network job in flight, their starts staggered:
    python tools/probes/pipelined_jobs.py [--job vgg16] [--jobs 12] "instances:offset_ms" ...
Each instance runs its jobs back to back on its own streams / contexts / host threads; instance k starts k * offset_ms after
instance 0.  Prints layers/s per setting (every job does all of its work; masks compared with a job run alone)."""
import argparse, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from benchkit import common as _c   # noqa
from benchkit.common import CD_FLAGS, cpjobs
from cpmi355 import shard


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--job", default="vgg16")
    ap.add_argument("--jobs", type=int, default=12)
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    specs = cpjobs.JOBS[a.job]()
    per_stream = 1 if a.job != "resnet50" else {"default": 2, 2048: 1}
    data = {s["layer_id"]: cpjobs.synth(s)[:3] for s in specs}
    max_inst = max(int(s.split(":")[0]) for s in a.settings)
    sets = [shard.ResidentLayerSet(0, specs, lambda sp: data[sp["layer_id"]], per_stream=per_stream, flags=CD_FLAGS, borrow_results=True)
            for _ in range(max_inst)]
    ref = [np.array(r[0]) for r in sets[0]()]
    for rs in sets:
        rs()
        rs()
    for setting in a.settings:
        n, off = setting.split(":")
        n, off = int(n), float(off)
        outs = [None] * n

        def loop(k):
            if k and off > 0:
                time.sleep(k * off * 1e-3)
            for _ in range(a.jobs):
                outs[k] = sets[k]()

        th = [threading.Thread(target=loop, args=(k,)) for k in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        ok = all(np.array_equal(np.array(r[0]), m) for o in outs for r, m in zip(o, ref))
        print("%d instance(s), start offset %.1f ms: %.1f layers/s, %.2f ms per job, masks %s" % (
            n, off, len(specs) * n * a.jobs / el, el / (n * a.jobs) * 1e3, "ok" if ok else "DIFFER"), flush=True)
    for rs in sets:
        rs.close()


if __name__ == "__main__":
    main()
