"""Row-sharded pruning of ONE large layer over the GPUs of a node (secondary sharding).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 \\
        tools/rowshard_bench.py --N 20000 --c 512 --n 512 --rank 276

Every rank holds N / world rows (synthetic, seeded per rank), resident in HBM before the timed calls; prints the
per-phase seconds of rank 0 (max over ranks for the total).  CP_BENCH_DIST_BACKEND=gloo runs several ranks on one
GPU (flow check only: the all-reduces then go through the host)."""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "channel-pruning_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=20000)
    ap.add_argument("--c", type=int, default=512)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--rank", type=int, default=276)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch                      # before the library: torch's HIP runtime serves both (capi.load)
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("CP_BENCH_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    d = None
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        d = dist
    from cpmi355 import capi
    from cpmi355.shard import RowShardEngine, prune_layer_rows, row_range
    lo, hi = row_range(a.N, world, rank)
    rs = np.random.RandomState(77)
    W2 = (rs.randn(a.n, a.c, a.k, a.k) * 0.05).astype(np.float32)
    rs = np.random.RandomState(1000 + rank)
    X = np.maximum(rs.randn(hi - lo, a.c, a.k, a.k), 0).astype(np.float32)
    Y = X.reshape(hi - lo, -1).astype(np.float64) @ W2.reshape(a.n, -1).T.astype(np.float64) + 0.01 * rs.randn(hi - lo, a.n)
    ctx = capi.Context(local)
    eng = RowShardEngine(ctx, flags=capi.CP_CD_RECIPROCAL | capi.CP_CD_DELTA)
    eng.load_rows(X, Y)
    out = None
    for rep in range(a.reps + 1):               # first repetition = warm-up
        tm = {}
        if d is not None:
            d.barrier()
        t0 = time.perf_counter()
        idxs, W, b, alpha = prune_layer_rows(eng, X, W2, Y, lo, a.N, a.rank, 1e-3, dist=d,
                                             rng=np.random.RandomState(4321), timings=tm)
        total = time.perf_counter() - t0
        t = torch.tensor([total], dtype=torch.float64)
        if d is not None:
            t = t.cuda() if backend == "nccl" else t
            d.all_reduce(t, op=dist.ReduceOp.MAX)
        out = dict(world=world, backend=backend if world > 1 else "none", N=a.N, c=a.c, n=a.n, k=a.k, kept=int(idxs.sum()),
                   fits=len(eng.fits), seconds_total_max_over_ranks=float(t.item()),
                   seconds_by_phase_rank0={k_: round(v, 6) for k_, v in tm.items()})
    if rank == 0:
        print(json.dumps(out))
    eng.free()
    ctx.close()
    if d is not None:
        d.destroy_process_group()


if __name__ == "__main__":
    main()
