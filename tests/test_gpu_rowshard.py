"""GPU: one layer with its rows spread over the ranks (secondary sharding) against the
reference's golden vectors.  World size 2 runs as two processes on the one GPU of the box with the "gloo" backend
(RCCL refuses two ranks on one device; on a multi-GPU node the same code reduces device tensors with "nccl")."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT

pytestmark = pytest.mark.gpu

REL_W = 1e-5


def _load(name):
    import cp_oracle
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    p = json.loads(str(g["params"]))
    X, W2, Y, B2 = cp_oracle.synth_layer(p["layer_id"], p["N"], p["c"], p["n"], p["k"], dead=p.get("dead", 0),
                                         residual=p.get("residual", False))
    return g, p, X.astype(np.float64), W2, Y


def _relfro(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / nb if nb > 0 else np.linalg.norm(a - b)


def _run(name, dist, rank, world):
    from cpmi355 import capi
    from cpmi355.shard import RowShardEngine, prune_layer_rows, row_range
    g, p, X, W2, Y = _load(name)
    lo, hi = row_range(p["N"], world, rank)
    ctx = capi.Context(0)
    eng = RowShardEngine(ctx, flags=capi.CP_CD_RECIPROCAL | capi.CP_CD_DELTA)
    rng = np.random.RandomState(0)
    rng.seed(1234 + p["layer_id"])
    idxs, W, b, alpha = prune_layer_rows(eng, X[lo:hi], W2, Y[lo:hi], lo, p["N"], p["rank"], p.get("alpha_in", 1e-3),
                                         dist=dist, rank_tol=p.get("rank_tol", .1), rng=rng,
                                         ridge=float(p.get("fc_ridge", 0)))
    out = dict(mask_equal=bool(np.array_equal(idxs, g["idxs"])), alpha_equal=bool(alpha == float(g["alpha_out"])),
               fits_equal=bool(p["rank"] == p["c"] or
                               np.array_equal(np.array(eng.fits, dtype=np.float64).reshape(-1, 3), g["fits"])),
               rng_equal=bool(int(rng.randint(0, 2147483647)) == int(g["rng_next"])),
               shape_equal=bool(W.shape == g["newW2"].shape),
               errW=float(_relfro(W, g["newW2"])) if W.shape == g["newW2"].shape else 1.0,
               errb=float(_relfro(b, g["newB2"])), fallback=int(eng.refit_info.fallback),
               digest=float(np.abs(W).sum()))
    ctx.close()
    return out


def _worker(rank, world, port, names, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for sub in ("channel-pruning_amd", "oracle"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    import torch  # noqa: F401  -- before the first Context: torch's HIP runtime has to be the one in the process
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, {name: _run(name, dist, rank, world) for name in names}))
    finally:
        if dist is not None:
            dist.destroy_process_group()


def _launch(world, names):
    """fresh processes (the pytest process already runs the library on the system HIP runtime)"""
    import multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29600 + (os.getpid() % 2000) + world
    procs = [mpc.Process(target=_worker, args=(r, world, port, names, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


CASES = ["s01_c32_k3", "s05_rank_eq_c", "s07_N_lt_p", "s09_ridge", "s08_resid_k1", "s11_rank_eq_c_dead",
         "L02_conv3_1_conv3_2"]


def _check(name, r):
    assert r["mask_equal"], name + ": channel mask differs from the reference"
    assert r["fits_equal"] and r["alpha_equal"] and r["rng_equal"], name + ": search log / alpha / RNG stream differ"
    assert r["shape_equal"] and r["errW"] <= REL_W and r["errb"] <= REL_W, "%s: W %.2e b %.2e" % (name, r["errW"], r["errb"])


def test_row_sharded_single_rank_matches_reference_golden():
    """world size 1 (no collective): the three-call refit equals the reference like the fused path does."""
    got = _launch(1, CASES)[0]
    for name in CASES:
        _check(name, got[name])
    assert got["s07_N_lt_p"]["fallback"] == 1     # N < p: minimum-norm branch through the shard tail
    assert got["s01_c32_k3"]["fallback"] == 0


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_world_matches_reference_golden(world):
    """Rows split over `world` ranks: masks, per-fit logs, alpha and the RNG stream identical to the reference on
    every rank; weights within 1e-5 (the Gram is a sum of per-rank partials); all ranks hold the same result."""
    got = _launch(world, CASES)
    for name in CASES:
        for r in range(world):
            _check(name, got[r][name])
        assert len({got[r][name]["digest"] for r in range(world)}) == 1, name + ": ranks disagree"


def _assist_worker(rank, world, port, names, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for sub in ("channel-pruning_amd", "oracle"):
        sys.path.insert(0, os.path.join(ROOT, sub))
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpmi355 import capi
    from cpmi355.shard import RowShardEngine, prune_layer_assisted
    group = dist.new_group(ranks=[0, 1])
    out = {}
    try:
        for name in names:
            g, p, X, W2, Y = _load(name)
            cut = p["N"] // 2
            ctx = capi.Context(0)
            eng = RowShardEngine(ctx, flags=0)
            if rank == 0:
                eng.load_rows(X[:cut], Y[:cut])
                rng = np.random.RandomState(1234 + p["layer_id"])
                idxs, W, b, alpha = prune_layer_assisted(eng, "owner", dist, group, 0, p["c"], W2, p["N"], p["rank"],
                                                         p.get("alpha_in", 1e-3), X=X, Y=Y, rank_tol=p.get("rank_tol", .1), rng=rng)
                out[name] = dict(mask_equal=bool(np.array_equal(idxs, g["idxs"])), alpha_equal=bool(alpha == float(g["alpha_out"])),
                                 fits_equal=bool(np.array_equal(np.array(eng.fits, dtype=np.float64).reshape(-1, 3), g["fits"])),
                                 rng_equal=bool(int(rng.randint(0, 2147483647)) == int(g["rng_next"])),
                                 shape_equal=bool(W.shape == g["newW2"].shape),
                                 errW=float(_relfro(W, g["newW2"])) if W.shape == g["newW2"].shape else 1.0,
                                 errb=float(_relfro(b, g["newB2"])))
            else:
                eng.load_rows(X[cut:], Y[cut:])
                assert prune_layer_assisted(eng, "helper", dist, group, 0, p["c"], W2, p["N"], p["rank"], 1e-3) is None
            eng.free()
            ctx.close()
        q.put((rank, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_layer_whose_owner_is_helped_by_a_second_rank_matches_reference_golden():
    """prune_layer_assisted on the GPU: the owner (rank 0) searches alone and broadcasts the mask, the helper (rank 1, the other
    half of the rows; both on this one GPU, gloo) contributes its column sums and its share of the normal equations, the owner
    solves: masks, per-fit logs, alpha and the RNG stream identical to the reference, weights within 1e-5."""
    import multiprocessing as mp
    names = ["s01_c32_k3", "s08_resid_k1", "L02_conv3_1_conv3_2"]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 29300 + (os.getpid() % 2000)
    procs = [mpc.Process(target=_assist_worker, args=(r, 2, port, names, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for name in names:
        _check(name, got[0][name])
    assert got[1] == {}


def test_row_shard_engine_explains_the_load_order(ctx):
    """In THIS process the library came first (session fixture), so torch.cuda cannot start: the engine says why."""
    from cpmi355.shard import RowShardEngine
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("torch.cuda already initialised in this process")
    with pytest.raises(RuntimeError, match="before creating the first cpmi355 Context"):
        RowShardEngine(ctx)
