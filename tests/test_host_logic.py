"""CPU: host-side logic of the product (no GPU compute): config mirror, Worker, layer sharding
(world_size-2 gloo), RNG-stream bookkeeping of the alpha search, loud failure without a GPU."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN_DIR as _GOLDEN, ROOT


def test_cfgs_mirror_has_the_keys_the_path_reads():
    import lib.cfgs as cfgs
    from lib.cfgs import c as dcfgs
    assert cfgs.alpha == 1e-3                                  # cfgs.py:18
    assert dcfgs.dic.rank_tol == .1 and dcfgs.dic.keep == 3.   # cfgs.py:74-75
    assert dcfgs.nBatches == 500 and dcfgs.nPointsPerLayer == 10
    assert dcfgs.solver == cfgs.solvers.sk and dcfgs.ls == 'linear' and dcfgs.fc_ridge == 0
    cfgs.set_nBatches(7)
    assert dcfgs.nBatches == 7 and dcfgs.nBatches_fc == 7
    cfgs.set_nBatches(500)


def test_worker_runs_target_in_a_child_and_returns_its_dict():
    from lib.worker import Worker

    def target(a, b, queue=None):
        return {"sum": a + b, "pid": os.getpid()}

    w = Worker()
    out = w.do(target, a=2, b=3)
    assert out["sum"] == 5 and out["pid"] != os.getpid()
    out = w.do(a=10, b=1)                                      # cached target (worker.py:13-18)
    assert out["sum"] == 11


def test_worker_propagates_child_failures_instead_of_hanging():
    from lib.worker import Worker

    def bad(queue=None):
        raise ValueError("boom")

    with pytest.raises(RuntimeError, match="boom"):
        Worker().do(bad)

    def not_a_dict(queue=None):
        return 3

    with pytest.raises(RuntimeError):
        Worker().do(not_a_dict)


def test_dictionary_without_gpu_fails_loudly():
    """The product never falls back to a CPU path: no GPU -> CpError from context creation."""
    from cpmi355 import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    import lib.decompose as D
    rs = np.random.RandomState(0)
    X = rs.randn(100, 8, 3, 3)
    W2 = rs.randn(4, 8, 3, 3).astype(np.float32)
    with pytest.raises(capi.CpError):
        D.dictionary(X, W2, rs.randn(100, 4), rank=4)
    with pytest.raises(capi.CpError):
        D.fc_kernel(rs.randn(50, 6), rs.randn(50, 2))


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under channel-pruning_amd/ may reference it."""
    pkg = os.path.join(ROOT, "channel-pruning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "cp_oracle" not in text and "import oracle" not in text and "libcporacle" not in text, f


def test_lpt_assignment_balances_layers():
    from cpmi355.shard import assign_layers, layer_cost
    vgg = [(64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 512), (512, 512),
           (512, 512), (512, 512), (512, 512), (512, 512)]
    costs = [layer_cost(5000, c, n, 3, c // 2) for c, n in vgg]
    for world in (1, 2, 4, 8):
        owner = assign_layers(costs, world)
        assert set(owner) <= set(range(world)) and len(owner) == len(costs)
        load = [sum(cst for cst, o in zip(costs, owner) if o == r) for r in range(world)]
        assert max(load) <= sum(costs) / world + max(costs)   # LPT bound
    assert assign_layers(costs, 2) == assign_layers(costs, 2)  # deterministic


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from cpmi355.shard import prune_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [dict(layer_id=i + 1, N=300, c=c, n=n, k=k, rank=r)
             for i, (c, n, k, r) in enumerate([(16, 12, 3, 8), (24, 16, 3, 12), (32, 8, 1, 10), (16, 16, 3, 16)])]
    calls = []

    def compute(s):   # the checker stands in for the GPU here (CPU test of the sharding logic only)
        calls.append(s["layer_id"])
        X, W2, Y, B2 = cp_oracle.synth_layer(s["layer_id"], s["N"], s["c"], s["n"], s["k"])
        rng = np.random.RandomState(1234 + s["layer_id"])
        out = cp_oracle.dictionary_oracle(X.astype(np.float64), W2, Y, s["rank"], B2, rng=rng, lasso="c_gram",
                                          ls="numpy")
        return out[0], out[1], out[2]

    res = prune_sharded(specs, compute, dist=dist, exchange="allgather")
    import hashlib   # exact bytes: a floating-point sum would depend on the alignment of the (sliced) receive buffer
    q.put((rank, calls, [(r[0].tolist(), r[1].shape, hashlib.sha1(np.ascontiguousarray(r[1]).tobytes()).hexdigest(),
                          hashlib.sha1(np.ascontiguousarray(r[2]).tobytes()).hexdigest()) for r in res]))
    dist.destroy_process_group()


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hashlib

    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [dict(layer_id=i, N=10, c=c, n=n, k=k, rank=1) for i, (c, n, k) in enumerate([(16, 3, 3), (5, 4, 1), (9, 2, 3), (16, 1, 1)])]
    owner = [2, 0, 2, 2]            # rank 1 owns nothing, rank 2 three layers of different shapes
    mine = {}
    for i, s in enumerate(specs):
        if owner[i] == rank:
            rs = np.random.RandomState(50 + i)
            idxs = rs.rand(s["c"]) < 0.6
            idxs[0] = True
            mine[i] = (idxs, rs.randn(s["n"], int(idxs.sum()), s["k"], s["k"]), rs.randn(s["n"]))
    res = shard.exchange_results(specs, owner, mine, dist, mode="allgather")
    q.put((rank, [(r[0].tolist(), r[1].shape, hashlib.sha1(np.ascontiguousarray(r[1]).tobytes()).hexdigest(),
                   hashlib.sha1(np.ascontiguousarray(r[2]).tobytes()).hexdigest()) for r in res],
           dict(shard.LAST_EXCHANGE_MS)))
    dist.destroy_process_group()


def test_exchange_results_three_ranks_one_of_them_empty():
    """exchange_results with uneven segments: a rank that owns nothing (empty segment), one that owns three layers of
    different shapes; every rank ends with the same bytes for every layer, and the segment lay-out is result_segments'."""
    import hashlib
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_exchange_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == got[2][1]
    for i, (m, shape, hw, hb) in enumerate(got[0][1]):      # equal to what the owner produced
        rs = np.random.RandomState(50 + i)
        c, n, k = [(16, 3, 3), (5, 4, 1), (9, 2, 3), (16, 1, 1)][i]
        idxs = rs.rand(c) < 0.6
        idxs[0] = True
        W, b = rs.randn(n, int(idxs.sum()), k, k), rs.randn(n)
        assert m == idxs.tolist() and tuple(shape) == W.shape
        assert hw == hashlib.sha1(W.tobytes()).hexdigest() and hb == hashlib.sha1(b.tobytes()).hexdigest()
    assert got[1][2]["bytes_sent"] == 0 and got[2][2]["bytes_received"] == got[0][2]["bytes_sent"]
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import shard
    masks = [np.array(m) for m, _, _, _ in got[0][1]]
    specs = [dict(n=n, k=k) for (c, n, k) in [(16, 3, 3), (5, 4, 1), (9, 2, 3), (16, 1, 1)]]
    shapes, offs, seg = shard.result_segments(specs, [2, 0, 2, 2], masks, 3)
    assert seg[1] == 0 and offs[0] == 0 and offs[2] == int(np.prod(shapes[0])) + 3 and seg[0] == int(np.prod(shapes[1])) + 4


def _vgg16_world8_specs():
    """the vgg16 job's layer table with the costs bench.py plans with, channel counts scaled down 8x (the exchange logic sees
    the real owner table and segment structure; the payload stays small)"""
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import jobs
    cost = {64: 1.5, 128: 3.0, 256: 6.8, 512: 15.5}      # bench.py::VGG16_COST_MS
    return [dict(layer_id=s["layer_id"], name=s["name"], N=40, c=s["c"] // 8, n=s["n"] // 8, k=s["k"], rank=s["rank"] // 8,
                 cost=cost[s["c"]]) for s in jobs.vgg16_4x()]


def _exchange_modes_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hashlib

    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [dict(layer_id=i, N=10, c=c, n=n, k=k, rank=1) for i, (c, n, k) in enumerate([(16, 3, 3), (5, 4, 1), (9, 2, 3), (16, 1, 1)])]
    owner = [2, 0, 2, 1]
    mine = {}
    for i, s in enumerate(specs):
        if owner[i] == rank:
            rs = np.random.RandomState(50 + i)
            idxs = rs.rand(s["c"]) < 0.6
            idxs[0] = True
            mine[i] = (idxs, rs.randn(s["n"], int(idxs.sum()), s["k"], s["k"]), rs.randn(s["n"]))
    sha = lambda a: None if a is None else hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()   # noqa: E731
    out = {}
    for mode in ("allgather", "gather", "masks"):
        res = shard.exchange_results(specs, owner, mine, dist, mode=mode)
        out[mode] = ([(r[0].tolist(), sha(r[1]), sha(r[2])) for r in res], dict(shard.LAST_EXCHANGE_MS))
    res = shard.prune_sharded(specs, compute_fn=lambda s_: mine[s_["layer_id"]], dist=dist, owner=owner)     # the default
    out["default"] = ([(r[0].tolist(), sha(r[1]), sha(r[2])) for r in res], dict(shard.LAST_EXCHANGE_MS))
    q.put((rank, out))
    dist.destroy_process_group()


def test_exchange_modes_gather_to_rank0_is_the_default_world_size_3_gloo():
    """north_star's split of the exchange: the masks' all_gather reaches every rank in every mode; the packed (W, b) go to rank 0
    only by default ("gather": exact segment lengths, point to point), everywhere with "allgather", nowhere with "masks".
    Rank 0's results are byte-identical in "gather" and "allgather"; the other ranks keep their own layers' weights and see
    (mask, None, None) for foreign ones; bytes sent per rank: its own segment once (gather) against world - 1 times over its links."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33100 + (os.getpid() % 1900)
    procs = [ctx.Process(target=_exchange_modes_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owner = [2, 0, 2, 1]
    full = got[0]["allgather"][0]
    assert all(w is not None and b is not None for _, w, b in full)
    for r in range(3):
        assert got[r]["allgather"][0] == full
        for mode in ("gather", "masks", "default"):
            res, info = got[r][mode]
            assert [m for m, _, _ in res] == [m for m, _, _ in full]                     # every mask on every rank
            for i, (m, w, b) in enumerate(res):
                has = owner[i] == r or (mode != "masks" and r == 0)
                assert ((w, b) == full[i][1:]) if has else (w is None and b is None)
        assert got[r]["default"][1]["mode"] == "gather"
        seg = got[r]["allgather"][1]["bytes_sent"]
        assert got[r]["allgather"][1]["link_bytes_out"] == 2 * seg
        assert got[r]["gather"][1]["bytes_sent"] == got[r]["gather"][1]["link_bytes_out"] == (0 if r == 0 else seg)
        assert got[r]["masks"][1]["bytes_sent"] == 0
    assert got[0]["gather"][1]["bytes_received"] == got[1]["gather"][1]["bytes_sent"] + got[2]["gather"][1]["bytes_sent"]
    assert got[1]["gather"][1]["bytes_received"] == 0


def _exchange8_worker(rank, world, port, q, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hashlib

    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = _vgg16_world8_specs()
    if case == "five_layers":
        specs = specs[7:]                 # the five widest layers alone: three of the eight ranks own nothing
    owner = shard.plan_owners(specs, world)
    mine = {}
    for i, s in enumerate(specs):
        if owner[i] == rank:
            rs = np.random.RandomState(70 + i)
            idxs = rs.rand(s["c"]) < 0.87
            idxs[0] = True
            mine[i] = (idxs, rs.randn(s["n"], int(idxs.sum()), s["k"], s["k"]), rs.randn(s["n"]))
    res = shard.exchange_results(specs, owner, mine, dist, mode="allgather")
    q.put((rank, owner, [(r[0].tolist(), tuple(r[1].shape), hashlib.sha1(np.ascontiguousarray(r[1]).tobytes()).hexdigest(),
                          hashlib.sha1(np.ascontiguousarray(r[2]).tobytes()).hexdigest()) for r in res],
           dict(shard.LAST_EXCHANGE_MS)))
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["vgg16_job", "five_layers"])
def test_exchange_results_world_size_8_with_the_vgg16_owner_table(case):
    """A dry run of the 8-GPU exchange on CPU (gloo, eight processes): the vgg16 job's owner table as bench.py plans it (LPT
    over the measured costs: uneven segments, ranks with two or three layers), and the five widest layers alone (three ranks
    own nothing: empty segments).  Every rank ends with every layer's bytes exactly as its owner produced them."""
    import hashlib
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (7 if case == "five_layers" else 0)
    procs = [ctx.Process(target=_exchange8_worker, args=(r, 8, port, q, case)) for r in range(8)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owner = got[0][1]
    assert all(g[1] == owner for g in got) and all(g[2] == got[0][2] for g in got)
    specs = _vgg16_world8_specs()
    specs = specs[7:] if case == "five_layers" else specs
    loads = [sum(1 for o in owner if o == r) for r in range(8)]
    if case == "vgg16_job":
        assert min(loads) >= 1 and max(loads) >= 2 and sorted(owner[7:]) == sorted(set(owner[7:]))   # a GPU per 512-channel layer
    else:
        assert loads.count(0) == 3 and max(loads) == 1
    for i, (m, shape, hw, hb) in enumerate(got[0][2]):
        s = specs[i]
        rs = np.random.RandomState(70 + i)
        idxs = rs.rand(s["c"]) < 0.87
        idxs[0] = True
        W, b = rs.randn(s["n"], int(idxs.sum()), s["k"], s["k"]), rs.randn(s["n"])
        assert m == idxs.tolist() and shape == W.shape
        assert hw == hashlib.sha1(W.tobytes()).hexdigest() and hb == hashlib.sha1(b.tobytes()).hexdigest()
    sent = [g[3]["bytes_sent"] for g in got]
    assert sum(sent) == sum(g[3]["bytes_received"] for g in got) // 7
    assert all((sent[r] == 0) == (loads[r] == 0) for r in range(8))


def _rounds_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import hashlib
    import time

    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # six layers: two heavy (0.6 s of "pruning" each), four light (0.05 s); costs in the specs, as bench.py carries them
    shapes = [(24, 6, 3, 0.6), (8, 4, 1, 0.05), (12, 5, 3, 0.05), (24, 6, 3, 0.6), (8, 3, 3, 0.05), (10, 4, 1, 0.05)]
    specs = [dict(layer_id=i, N=10, c=c, n=n, k=k, rank=1, cost=cost) for i, (c, n, k, cost) in enumerate(shapes)]
    owner = shard.plan_owners(specs, world)
    rounds = shard.plan_rounds(specs)
    ended = {}

    def compute(s):
        time.sleep(s["cost"])
        rs = np.random.RandomState(90 + s["layer_id"])
        idxs = rs.rand(s["c"]) < 0.7
        idxs[0] = True
        ended[s["layer_id"]] = time.perf_counter()
        return idxs, rs.randn(s["n"], int(idxs.sum()), s["k"], s["k"]), rs.randn(s["n"])

    own = [s for s, o in zip(specs, owner) if o == rank]
    exchanged = []
    real = shard.exchange_results

    def logged(*a, **kw):
        out = real(*a, **kw)
        exchanged.append(time.perf_counter())
        return out

    shard.exchange_results = logged
    res = shard.prune_sharded(specs, dist=dist, owner=owner, compute_many=shard.ThreadedLayerSet(own, compute), rounds=rounds,
                              exchange="allgather")
    shard.exchange_results = real
    plain = shard.prune_sharded(specs, dist=dist, owner=owner, compute_many=shard.ThreadedLayerSet(own, compute), exchange="allgather")
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(res, plain))
    heavy_end = max([ended[s["layer_id"]] for s in own if s["cost"] > 0.3], default=None)
    q.put((rank, owner, rounds, same, len(exchanged), (exchanged[0] < heavy_end) if heavy_end is not None else None,
           [hashlib.sha1(np.ascontiguousarray(r[1]).tobytes()).hexdigest() for r in res], list(shard.LAST_EXCHANGE_MS.get("rounds", []))))
    dist.destroy_process_group()


def test_exchange_in_rounds_overlaps_the_heavy_layers_world_size_3_gloo():
    """prune_sharded(rounds=plan_rounds(...)): the results of the light layers are exchanged (mask all_gather + packed
    all_gather, as always) while the heavy layers are still being pruned, the heavy layers' afterwards.  Three ranks, six
    layers (two heavy): every rank ends with exactly what the one-exchange path gives, two exchanges ran, and on the ranks
    that own a heavy layer the first exchange was over before that layer finished."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rounds_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][2] == [1, 0, 0, 1, 0, 0]
    assert all(g[3] for g in got) and all(g[4] == 2 for g in got) and all(len(g[7]) == 2 for g in got)
    assert got[0][6] == got[1][6] == got[2][6]
    overl = [g[5] for g in got if g[5] is not None]
    assert len(overl) == 2 and all(overl)          # the two heavy layers sit on two different ranks (LPT)


def test_sharded_pruning_world_size_2_gloo():
    """Two ranks split four layers, then every rank holds every layer's (mask, W, b); the union of
    the work is exactly one call per layer and the results equal a single-process run."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got.sort()
    calls = sorted(got[0][1] + got[1][1])
    assert calls == [1, 2, 3, 4] and got[0][1] and got[1][1]
    assert got[0][2] == got[1][2]
    # single-process reference of the same sharding code path
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from cpmi355.shard import prune_sharded
    specs = [dict(layer_id=i + 1, N=300, c=c, n=n, k=k, rank=r)
             for i, (c, n, k, r) in enumerate([(16, 12, 3, 8), (24, 16, 3, 12), (32, 8, 1, 10), (16, 16, 3, 16)])]

    def compute(s):
        X, W2, Y, B2 = cp_oracle.synth_layer(s["layer_id"], s["N"], s["c"], s["n"], s["k"])
        out = cp_oracle.dictionary_oracle(X.astype(np.float64), W2, Y, s["rank"], B2,
                                          rng=np.random.RandomState(1234 + s["layer_id"]), lasso="c_gram", ls="numpy")
        return out[0], out[1], out[2]

    single = prune_sharded(specs, compute)
    import hashlib
    for (m, shape, hw, hb), r in zip(got[0][2], single):
        assert m == r[0].tolist() and tuple(shape) == r[1].shape
        assert hw == hashlib.sha1(np.ascontiguousarray(r[1]).tobytes()).hexdigest()
        assert hb == hashlib.sha1(np.ascontiguousarray(r[2]).tobytes()).hexdigest()


class _NumpyRowEngine:
    """CPU stand-in for cpmi355.shard.RowShardEngine (same methods, NumPy/oracle arithmetic, unpadded buffers):
    lets the world-size-2 gloo test drive the real exchange logic of prune_layer_rows without a GPU."""

    def buffer(self, elems):
        import torch
        return torch.zeros(int(elems), dtype=torch.float64)

    def select(self, Xs, W2, Ys, rank, alpha_in, rank_tol, rng):
        import cp_oracle

        class FirstDrawIsIdentity:      # dictionary_oracle draws its sample subset first: hand it every row
            def __init__(self):
                self.first = True

            def randint(self, lo, hi, size=None):
                if self.first:
                    self.first = False
                    return np.arange(Xs.shape[0])
                return rng.randint(lo, hi, size)

        out = cp_oracle.dictionary_oracle(Xs.astype(np.float64), W2, Ys, rank, alpha_in=alpha_in, rank_tol=rank_tol,
                                          rng=FirstDrawIsIdentity(), lasso="c_gram", refit="none")
        return out[0], out[3]

    def load_rows(self, X_local, Y_local):
        self.X = X_local.reshape(X_local.shape[0], X_local.shape[1], -1).astype(np.float64)
        self.Y = Y_local
        self.kk, self.n = self.X.shape[2], Y_local.shape[1]

    def layout(self, kept):
        p = kept * self.kk
        return p + self.n, p * (p + self.n)

    def _cols(self, mask):
        return self.X[:, mask.astype(bool)].reshape(self.X.shape[0], -1)

    def sums(self, mask, sums):
        sums[:] = __import__("torch").from_numpy(np.concatenate([self._cols(mask).sum(0), self.Y.sum(0)]))

    def gram(self, mask, N_total, sums, gram):
        import torch
        Xk = self._cols(mask)
        p = Xk.shape[1]
        m = sums.numpy() / N_total
        Xc, Yc = Xk - m[:p], self.Y - m[p:]
        gram[:] = torch.from_numpy(np.concatenate([(Xc.T @ Xc).ravel(), (Xc.T @ Yc).ravel()]))

    def solve(self, kept, N_total, ridge, sums, gram):
        p = kept * self.kk
        g, m = gram.numpy(), sums.numpy() / N_total
        G, R = g[:p * p].reshape(p, p), g[p * p:].reshape(p, self.n)
        W = np.linalg.lstsq(G + ridge * np.eye(p), R, rcond=None)[0].T
        return W, m[p:] - W @ m[:p]

    def free(self):
        pass


def _row_shard_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from cpmi355.shard import prune_layer_rows, row_range
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    for lid, (N, c, n, k, r) in enumerate([(600, 16, 12, 3, 8), (500, 24, 16, 1, 10), (400, 12, 12, 3, 12)], start=1):
        X, W2, Y, B2 = cp_oracle.synth_layer(lid, N, c, n, k)
        lo, hi = row_range(N, world, rank)
        rng = np.random.RandomState(1234 + lid)
        idxs, W, b, alpha = prune_layer_rows(_NumpyRowEngine(), X[lo:hi], W2, Y[lo:hi], lo, N, r, 1e-3, dist=dist,
                                             rng=rng)
        out.append((idxs.tolist(), W, b, alpha, int(rng.randint(0, 2147483647))))
    q.put((rank, out))
    dist.destroy_process_group()


def _assist_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from cpmi355.shard import prune_layer_assisted
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.new_group(ranks=[0, 2])              # every rank creates it; rank 1 stands by
    out = []
    for lid, (N, c, n, k, r) in enumerate([(600, 16, 12, 3, 8), (500, 24, 16, 1, 10), (400, 12, 12, 3, 12)], start=1):
        X, W2, Y, B2 = cp_oracle.synth_layer(lid, N, c, n, k)
        cut = N // 2 + 7                               # the owner keeps the first rows, the helper the rest (uneven on purpose)
        if rank == 0:
            eng = _NumpyRowEngine()
            eng.load_rows(X[:cut], Y[:cut])
            rng = np.random.RandomState(1234 + lid)
            idxs, W, b, alpha = prune_layer_assisted(eng, "owner", dist, group, 0, c, W2, N, r, 1e-3, X=X, Y=Y, rng=rng)
            out.append((idxs.tolist(), W, b, alpha, int(rng.randint(0, 2147483647))))
        elif rank == 2:
            eng = _NumpyRowEngine()
            eng.load_rows(X[cut:], Y[cut:])
            assert prune_layer_assisted(eng, "helper", dist, group, 0, c, W2, N, r, 1e-3) is None
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_layer_whose_owner_is_helped_by_a_second_rank_world_size_3_gloo():
    """prune_layer_assisted: the owner runs the alpha search alone and broadcasts the mask; owner and helper then sum the column
    sums and the normal equations of their rows (two all-reduces inside their own group, a third rank standing by); the owner
    solves.  Mask, alpha and the reference's RNG stream exactly as a single process gives them, W / b to 1e-9 (NumPy stand-in
    for the GPU engine: the exchange logic is what runs here)."""
    import multiprocessing as mp
    import cp_oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_assist_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[1] == [] and got[2] == [] and len(got[0]) == 3
    for lid, (N, c, n, k, r) in enumerate([(600, 16, 12, 3, 8), (500, 24, 16, 1, 10), (400, 12, 12, 3, 12)], start=1):
        X, W2, Y, B2 = cp_oracle.synth_layer(lid, N, c, n, k)
        rng = np.random.RandomState(1234 + lid)
        ref = cp_oracle.dictionary_oracle(X.astype(np.float64), W2, Y, r, B2, rng=rng, lasso="c_gram", ls="numpy")
        idxs, W, b, alpha, nxt = got[0][lid - 1]
        assert idxs == ref[0].tolist() and alpha == ref[3] and nxt == int(rng.randint(0, 2147483647))
        assert np.linalg.norm(W - ref[1]) <= 1e-9 * np.linalg.norm(ref[1]) and np.linalg.norm(b - ref[2]) <= 1e-9 * max(1.0, np.linalg.norm(ref[2]))


class _FailingEngine(_NumpyRowEngine):
    """_NumpyRowEngine that raises in one named stage"""

    def __init__(self, stage):
        self.stage = stage

    def select(self, *a, **k):
        if self.stage == "select":
            raise ValueError("boom in select")
        return super().select(*a, **k)

    def gram(self, *a, **k):
        if self.stage == "gram":
            raise ValueError("boom in gram")
        return super().gram(*a, **k)


def _assist_failure_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from cpmi355 import shard
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    group = dist.new_group(ranks=[0, 1])
    N, c, n, k, r = 600, 16, 12, 3, 8
    X, W2, Y, B2 = cp_oracle.synth_layer(1, N, c, n, k)
    cut = N // 2
    seen = []
    # (stage that fails on the owner, stage that fails on the helper): the rank at fault raises its own error, the other one
    # AssistPeerError, and nobody stays behind in a collective (the barrier after each case would time out otherwise)
    for own_stage, help_stage in (("select", None), (None, "gram"), ("gram", None), (None, None)):
        eng = _FailingEngine(own_stage if rank == 0 else help_stage)
        eng.load_rows(X[:cut], Y[:cut]) if rank == 0 else eng.load_rows(X[cut:], Y[cut:])
        try:
            if rank == 0:
                shard.prune_layer_assisted(eng, "owner", dist, group, 0, c, W2, N, r, 1e-3, X=X, Y=Y, rng=np.random.RandomState(1235))
            else:
                shard.prune_layer_assisted(eng, "helper", dist, group, 0, c, W2, N, r, 1e-3)
            seen.append("ok")
        except shard.AssistPeerError:
            seen.append("peer")
        except ValueError as e:
            seen.append(str(e))
        dist.barrier()
    # prune_sharded: rank 1's layer raises -> both ranks raise (rank 0 a ShardPeerError naming rank 1) instead of rank 0
    # waiting in the exchange for ever; a clean job right after still works (the process group is intact)
    specs = [dict(layer_id=10 + i, N=200, c=8, n=6, k=1, rank=4) for i in range(4)]

    def compute(spec, fail=False):
        if fail and spec["layer_id"] == 11:
            raise ValueError("boom in layer 11")
        m = np.zeros(spec["c"], dtype=bool)
        m[:4] = True
        return m, np.full((spec["n"], 4, 1, 1), float(spec["layer_id"])), np.zeros(spec["n"])

    owner = [0, 1, 0, 1]
    for rounds in (None, [0, 0, 1, 1]):
        layer_set = shard.ThreadedLayerSet([specs[i] for i in range(4) if owner[i] == rank], lambda s_: compute(s_, True))
        try:
            shard.prune_sharded(specs, compute_many=layer_set, dist=dist, owner=owner, rounds=rounds)
            seen.append("ok")
        except shard.ShardPeerError as e:
            seen.append("peer%s" % e.ranks)
        except ValueError as e:
            seen.append(str(e))
        dist.barrier()
    res = shard.prune_sharded(specs, compute_fn=compute, dist=dist, owner=owner, exchange="allgather")
    seen.append([float(r_[1].ravel()[0]) for r_ in res])
    q.put((rank, seen))
    dist.destroy_process_group()


def test_a_failing_rank_never_leaves_its_peers_waiting_in_a_collective_gloo():
    """advisor, round 5: no failure path across ranks.  An assisted layer whose owner fails in the alpha search, or whose
    owner / helper fails in the Gram stage, ends with the faulty rank raising its own error and its peer AssistPeerError --
    both having walked broadcast -> all-reduce -> all-reduce; a rank whose layer raises inside prune_sharded (single exchange
    and exchange in rounds) tells the others through the mask all_gather (ShardPeerError on them).  The process group stays
    usable: a clean job follows on the same group."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_assist_failure_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][:4] == ["boom in select", "peer", "boom in gram", "ok"]
    assert got[1][:4] == ["peer", "boom in gram", "peer", "ok"]
    assert got[0][4:6] == ["peer[1]", "peer[1]"] and got[1][4:6] == ["boom in layer 11", "boom in layer 11"]
    assert got[0][6] == got[1][6] == [10.0, 11.0, 12.0, 13.0]


def test_plan_assists_uses_ranks_with_slack_only():
    """plan_assists: a helper for the layers whose cost test pays (N = 20000), never at the 5000-sample job; a rank helps at most
    one layer, never its own, and only if its own load is at most half the helped layer's cost."""
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import jobs, shard
    s5 = jobs.vgg16_5x()
    owner = shard.plan_owners(s5, 8)
    a = shard.plan_assists(s5, owner, 8)
    assert a and all(s5[i]["c"] == 512 for i in a) and len(set(a.values())) == len(a)
    assert all(h != owner[i] for i, h in a.items())
    assert shard.plan_assists(jobs.vgg16_4x(), shard.plan_owners(jobs.vgg16_4x(), 8), 8) == {}
    assert shard.plan_assists(s5, [0] * len(s5), 1) == {}


def test_row_sharded_layer_world_size_2_gloo():
    """prune_layer_rows with the rows of each layer split over two ranks: the sampled-row exchange and the two
    all-reduces reproduce the single-process result (mask, alpha and RNG stream exactly; W, b to 1e-9) on both
    ranks.  The arithmetic is the NumPy stand-in; the GPU engine is covered by tests/test_gpu_rowshard.py."""
    import multiprocessing as mp
    import cp_oracle
    from cpmi355.shard import row_range
    assert [row_range(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_row_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=900) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for lid, (N, c, n, k, r) in enumerate([(600, 16, 12, 3, 8), (500, 24, 16, 1, 10), (400, 12, 12, 3, 12)], start=1):
        X, W2, Y, B2 = cp_oracle.synth_layer(lid, N, c, n, k)
        rng = np.random.RandomState(1234 + lid)
        ref = cp_oracle.dictionary_oracle(X.astype(np.float64), W2, Y, r, B2, rng=rng, lasso="c_gram", ls="numpy")
        nxt = int(rng.randint(0, 2147483647))
        for rank in range(2):
            idxs, W, b, alpha, rng_next = got[rank][lid - 1]
            assert idxs == ref[0].tolist() and rng_next == nxt
            assert r == c or alpha == ref[3]
            assert np.linalg.norm(W - ref[1]) <= 1e-9 * np.linalg.norm(ref[1])
            assert np.linalg.norm(b - ref[2]) <= 1e-9 * max(1.0, np.linalg.norm(ref[2]))


def test_alpha_search_rng_rewind_matches_reference_consumption():
    """The device-mode search pre-draws seeds, then rewinds and re-draws exactly F of them: the
    stream position afterwards must equal 1 + F draws (what the reference consumes)."""
    from cpmi355 import pruner

    class FakeCtx:
        def lasso_alpha_search(self, *a, **k):
            return 5, 0.01, [(0.01, 3, 4)] * 5      # pretend the search used 5 fits

    class P(pruner.LayerProblem):
        def __init__(self):
            self.ctx, self.c, self.n, self.S, self.flags = FakeCtx(), 8, 4, 10, 0
            self.Qd = self.qd = self.statsd = self.wd = None
            self.fits = []

        def reset_w(self):
            pass

    rng = np.random.RandomState(42)
    P().alpha_search(4, 1e-3, .1, rng, mode="device")
    after = rng.randint(0, 2147483647)
    ref = np.random.RandomState(42)
    for _ in range(5):
        ref.randint(0, pruner.RAND_R_MAX)
    assert after == ref.randint(0, 2147483647)


def test_batched_pruning_consumes_each_layers_rng_like_the_reference(monkeypatch):
    """prune_layers_batched (cp_prune_layers stubbed out): every layer's own generator ends 1 + F draws further (the
    sample subset, one seed per fit the search used), the seeds handed to the library are the next MAX_FITS draws of
    that generator, rank == c layers draw only their samples, and a layer whose search did not settle is rewound and
    replayed fit by fit."""
    from cpmi355 import capi, pruner
    seen = {}

    class Prob:
        def __init__(self, c, n, N, unsettled=False):
            self.ctx, self.c, self.n, self.N, self.k, self.kk = object(), c, n, N, 3, 9
            self.Xd = self.W2d = self.Yd = None
            self.x_dtype = self.w_dtype = 0
            self.flags, self.unsettled, self.replayed, self.fits = 0, unsettled, [], []

        def lasso_gram(self, samples):
            self.replayed.append(("gram", len(samples)))

        def alpha_search(self, rank, alpha_in, rank_tol, rng, mode):
            self.replayed.append(("search", mode))
            rng.randint(0, pruner.RAND_R_MAX)          # the replay used one fit
            return 0.5

        def mask(self):
            return np.arange(self.c) < 3

        def refit(self, idxs, ridge=0.0):
            return np.zeros((self.n, 3 * self.kk)), np.zeros(self.n)

    def fake_prune_layers(jobs):
        out = []
        for i, j in enumerate(jobs):
            seen[i] = (np.array(j["samples"]), np.array(j["seeds"]))
            r = capi.PruneResult()
            prob = probs[i]
            if prob.unsettled:
                r.fits_used = -1
                out.append((r, None, None, None))
                continue
            r.fits_used = 0 if j["rank"] == j["c"] else 3 + i
            r.p, r.alpha = 2 * j["kk"], 0.25
            mask = np.arange(j["c"]) < 2
            out.append((r, mask, np.zeros((j["n"], 2 * j["kk"])), np.zeros(j["n"])))
        return out

    monkeypatch.setattr(capi.Context, "prune_layers", staticmethod(fake_prune_layers))
    probs = [Prob(8, 4, 400), Prob(8, 4, 400), Prob(8, 4, 400, unsettled=True), Prob(8, 4, 400)]
    ranks = [4, 4, 4, 8]
    rngs = [np.random.RandomState(100 + i) for i in range(4)]
    res = pruner.prune_layers_batched(probs, ranks, [1e-3] * 4, rngs)
    for i in range(4):
        ref = np.random.RandomState(100 + i)
        samples = ref.randint(0, 400, 20)
        assert np.array_equal(seen[i][0], samples)
        if ranks[i] != 8:
            state = ref.get_state()
            seeds = np.array([ref.randint(0, pruner.RAND_R_MAX) for _ in range(pruner.MAX_FITS)], dtype=np.uint32)
            assert np.array_equal(seen[i][1], seeds)
            ref.set_state(state)
        used = {0: 3, 1: 4, 2: 1, 3: 0}[i]                    # layer 2: replayed on the host, one fit
        for _ in range(used):
            ref.randint(0, pruner.RAND_R_MAX)
        assert rngs[i].randint(0, 1 << 30) == ref.randint(0, 1 << 30), i
    assert probs[2].replayed == [("gram", 20), ("search", "host")] and res[2][3] == 0.5
    assert res[0][3] == 0.25 and res[3][3] == 1e-4              # rank == c: the alpha argument's default, no search
    assert res[0][1].shape == (4, 2, 3, 3) and res[2][1].shape == (4, 3, 3, 3)


def test_cheap_rng_bookkeeping_equals_numpy_semantics():
    """draw_seeds (one vectorised randint) leaves the values and the generator state that the same number of scalar
    rng.randint(0, 2147483647) calls leave -- the reference draws them one per Lasso.fit (_cd_fast.pyx:164) -- and
    rng_mark / rng_rewind restore a legacy generator exactly (also across a Mersenne-Twister block boundary), for
    RandomState objects and for the np.random module itself."""
    from cpmi355 import pruner
    for seed in (0, 7, 1234, 987654):
        a, b = np.random.RandomState(seed), np.random.RandomState(seed)
        a.randint(0, 10, size=590)
        b.randint(0, 10, size=590)                              # 64 more draws cross the 624-word block
        ref = np.array([a.randint(0, pruner.RAND_R_MAX) for _ in range(64)], dtype=np.uint32)
        mark = pruner.rng_mark(b)
        got = pruner.draw_seeds(b, 64)
        assert got.dtype == np.uint32 and np.array_equal(got, ref)
        assert a.randint(0, 1 << 30) == b.randint(0, 1 << 30)
        pruner.rng_rewind(b, mark)
        assert np.array_equal(pruner.draw_seeds(b, 64), ref)
        pruner.rng_rewind(b, mark)
        pruner.draw_seeds(b, 9)
        a2 = np.random.RandomState(seed)
        a2.randint(0, 10, size=590)
        for _ in range(9):
            a2.randint(0, pruner.RAND_R_MAX)
        assert a2.randint(0, 1 << 30) == b.randint(0, 1 << 30)
    np.random.seed(42)
    mark = pruner.rng_mark(np.random)
    x = np.random.randint(0, 1000, size=10)
    pruner.rng_rewind(np.random, mark)
    assert np.array_equal(np.random.randint(0, 1000, size=10), x)

    class Other:                                                 # not a Mersenne Twister: falls back to get/set_state
        def __init__(self):
            self.s = 0

        def get_state(self):
            return self.s

        def set_state(self, s):
            self.s = s
    o = Other()
    m = pruner.rng_mark(o)
    o.s = 5
    pruner.rng_rewind(o, m)
    assert o.s == 0


def test_torch_sequential_provider_is_live_and_wired_like_the_reference():
    """lib/provider.py: conv blobs are pre-ReLU, <conv>_relu / pool blobs feed the consumers, and the blobs follow the
    weights the net holds at call time (what Net.R3's decomposition steps rely on).  CPU only: no Net method that
    needs the GPU is called."""
    import torch
    import torch.nn.functional as F
    from lib.net import ConvSpec, Net
    from lib.provider import TorchSequentialProvider
    rs = np.random.RandomState(0)
    specs = [ConvSpec("conv1_1", rs.randn(4, 3, 3, 3) * .2, rs.randn(4) * .1, "data"),
             ConvSpec("conv1_2", rs.randn(6, 4, 3, 3) * .2, rs.randn(6) * .1, "conv1_1_relu"),
             ConvSpec("conv2_1", rs.randn(5, 6, 3, 3) * .2, rs.randn(5) * .1, "pool1")]
    data = [rs.randn(2, 3, 8, 8).astype(np.float32) for _ in range(2)]
    prov = TorchSequentialProvider(data, pools={"conv1_2": ("pool1", 2, 2)}, num_threads=1)
    net = Net(specs, prov, nBatches=2, nPointsPerLayer=2)
    assert net._live and len(prov) == 2
    blobs = net.forward(1)
    x = torch.from_numpy(data[1])
    y1 = F.conv2d(x, torch.from_numpy(specs[0].W), torch.from_numpy(specs[0].b), padding=1)
    y2 = F.conv2d(F.relu(y1), torch.from_numpy(specs[1].W), torch.from_numpy(specs[1].b), padding=1)
    p1 = F.max_pool2d(F.relu(y2), 2, 2)
    y3 = F.conv2d(p1, torch.from_numpy(specs[2].W), torch.from_numpy(specs[2].b), padding=1)
    assert np.array_equal(blobs["conv1_1"], y1.numpy()) and np.array_equal(blobs["conv1_2"], y2.numpy())
    assert np.array_equal(blobs["pool1"], p1.numpy()) and np.array_equal(blobs["conv2_1"], y3.numpy())
    assert blobs["conv2_1"].shape == (2, 5, 4, 4) and (blobs["conv1_1"] < 0).any()      # pre-ReLU responses
    # live: a weight change shows up in the next forward (the cache is dropped by set_param_data)
    net.set_param_data("conv1_2", np.zeros_like(specs[1].W))
    again = net.forward(1)
    assert np.array_equal(again["conv1_2"], np.broadcast_to(specs[1].b[None, :, None, None], (2, 6, 8, 8)))
    # features sampled through the net's own extraction code, frozen points reused for a second extraction
    np.random.seed(3)
    feats, points = net.extract_features(["conv1_1", "conv2_1"], save=1)
    assert feats["conv1_1"].shape == (2 * 2 * 2, 4) and feats["conv2_1"].shape == (8, 5)
    feats2, _ = net.extract_features(["conv1_1"], points_dict=points, save=1)
    assert np.array_equal(feats["conv1_1"], feats2["conv1_1"])


def test_gpu_layer_batches_groups_by_width_and_keeps_layer_order(monkeypatch):
    """cpmi355.shard.GpuLayerBatches (device calls stubbed out): layers are grouped by channel count, each group is
    cut into calls of at most max_batch layers on a context and its siblings, every layer gets its own seeded RNG and
    the results come back in the order of the specs."""
    from cpmi355 import pruner, shard
    calls, freed, closed = [], [], []

    class FakeCtx:
        def __init__(self, name):
            self.name = name

        def sibling(self):
            return FakeCtx(self.name + "'")

        def close(self):
            closed.append(self.name)

    class FakeProb:
        def __init__(self, ctx, X, W2, Y, flags=0):
            self.ctx, self.tag = ctx, X

        def free(self):
            freed.append(self.tag)

    def fake_batched(probs, ranks, alpha_ins, rngs, rank_tol=.1):
        calls.append(([p.tag for p in probs], list(ranks), [int(r.randint(0, 1 << 30)) for r in rngs], len({id(p.ctx) for p in probs})))
        return [(np.array([True]), np.full((1, 1, 1, 1), p.tag), np.zeros(1), 0.1 * p.tag) for p in probs]

    monkeypatch.setattr(pruner, "LayerProblem", FakeProb)
    monkeypatch.setattr(pruner, "prune_layers_batched", fake_batched)
    specs = [dict(layer_id=i, N=100, c=c, n=4, k=3, rank=2) for i, c in enumerate([32, 64, 32, 32, 64, 16, 32])]
    eng = shard.GpuLayerBatches(FakeCtx("root"), lambda s: (s["layer_id"], None, None), max_batch=3, flags=0)
    out = shard.prune_sharded(specs, compute_many=eng)
    assert [int(W[0, 0, 0, 0]) for _, W, _ in out] == list(range(7))              # layer order restored
    assert [c[0] for c in calls] == [[5], [0, 2, 3], [6], [1, 4]]                  # by width, at most 3 per call
    assert all(c[3] == len(c[0]) for c in calls)                                  # distinct contexts within a call
    for tags, _, draws, _ in calls:
        assert draws == [int(np.random.RandomState(1234 + t).randint(0, 1 << 30)) for t in tags]
    assert sorted(freed) == list(range(7)) and len(closed) == 2                    # 2 siblings for batches of 3
    assert eng.alphas[4] == pytest.approx(0.4)


def test_resident_layer_set_chunks_streams_and_latency_layers(monkeypatch):
    """cpmi355.shard.ResidentLayerSet (device calls stubbed out): equal-width layers are cut into chunks of `per_stream`, every
    chunk has its own context (= stream) + siblings and host thread, single-layer chunks go through prune_layer, the others
    through prune_layers_batched, only the `precompute_heaviest` heaviest single-layer chunks run in latency mode, every run
    restarts each layer's RNG stream, results come back in the order of the specs."""
    from cpmi355 import capi, pruner, shard
    created, calls = [], []

    class FakeCtx:
        def __init__(self, device=0, name=None, priority=None):
            self.name = name or "ctx%d" % len(created)
            created.append(self.name)

        def sibling(self):
            return FakeCtx(name=self.name + "'")

        def close(self):
            pass

    class FakeProb:
        def __init__(self, ctx, X, W2, Y, flags=0):
            self.ctx, self.tag, self.fits = ctx, X, []

        def free(self):
            pass

    def fake_single(prob, rank, alpha_in, rank_tol=.1, rng=None, mode="device", latency_mode=True, **kw):
        calls.append(("single", [prob.tag], latency_mode, [int(rng.randint(0, 1 << 30))]))
        return np.array([True]), np.full((1, 1, 1, 1), prob.tag), np.zeros(1), 0.5

    def fake_batched(probs, ranks, alpha_ins, rngs, rank_tol=.1):
        calls.append(("batch", [p.tag for p in probs], False, [int(r.randint(0, 1 << 30)) for r in rngs]))
        return [(np.array([True]), np.full((1, 1, 1, 1), p.tag), np.zeros(1), 0.5) for p in probs]

    monkeypatch.setattr(capi, "Context", FakeCtx)
    monkeypatch.setattr(pruner, "LayerProblem", FakeProb)
    monkeypatch.setattr(pruner, "prune_layer", fake_single)
    monkeypatch.setattr(pruner, "prune_layers_batched", fake_batched)
    widths = [64, 64, 128, 256, 256, 256, 512, 512]
    specs = [dict(layer_id=i, N=5000, c=c, n=c, k=3, rank=int(c / 1.15)) for i, c in enumerate(widths)]
    rset = shard.ResidentLayerSet(0, specs, lambda s: (s["layer_id"], None, None), per_stream=2, precompute_heaviest=1)
    try:
        assert [len(ch["members"]) for ch in rset.chunks] == [2, 2, 1, 1, 2]          # widest first: 512x2 | 256x2, 256 | 128 | 64x2
        assert len(created) == 5 + 3                                                  # a context per chunk + a sibling per extra layer
        for _ in range(2):                                                            # repeated runs replay the same RNG streams
            calls.clear()
            out = rset.run()
            assert [int(W[0, 0, 0, 0]) for _, W, _, _ in out] == list(range(8))
            singles = [c for c in calls if c[0] == "single"]
            assert sorted(c[1][0] for c in singles) == [2, 5] and sorted(tuple(c[1]) for c in calls if c[0] == "batch") == [(0, 1), (3, 4), (6, 7)]
            # in a job the heaviest single layer gets its normal equations precomputed ("gram"), not the factorisation too
            assert [c[2] for c in singles if c[1] == [5]] == ["gram"] and [c[2] for c in singles if c[1] == [2]] == [False]
            for _, tags, _, draws in calls:
                assert draws == [int(np.random.RandomState(1234 + t).randint(0, 1 << 30)) for t in tags]
        assert [(r["c"], len(r["layers"])) for r in rset.chunk_report()] == [(512, 2), (256, 2), (256, 1), (128, 1), (64, 2)]
    finally:
        rset.close()
    # per_stream by width: {c: layers per stream, "default": ...} -- wide layers alone, narrow ones batched
    rset = shard.ResidentLayerSet(0, specs, lambda s: (s["layer_id"], None, None), per_stream={512: 1, 64: 2, "default": 3},
                                  precompute_heaviest=0)
    try:
        assert [(r["c"], len(r["layers"])) for r in rset.chunk_report()] == [(512, 1), (512, 1), (256, 3), (128, 1), (64, 2)]
        out = rset.run()
        assert [int(W[0, 0, 0, 0]) for _, W, _, _ in out] == list(range(8))
    finally:
        rset.close()


def test_resident_layer_set_gives_the_precompute_to_the_longest_searches(monkeypatch):
    """cpmi355.shard.ResidentLayerSet (device calls stubbed out): which of the equally heavy single-layer chunks run with their
    normal equations precomputed under the search is re-decided after every run -- the `precompute_heaviest` layers whose
    searches took the most coordinate steps (sum of n_iter over the fits x channels; the later starter wins a tie); the
    first run takes the first ones; CP_PRECOMPUTE_ADAPT=0 keeps that choice; narrower layers never enter the pool."""
    from cpmi355 import capi, pruner, shard
    calls = []
    n_iter = {0: 10, 1: 30, 2: 10, 3: 20, 4: 99}          # layer_id -> epochs of its one fit

    class FakeCtx:
        def __init__(self, device=0, name=None, priority=None):
            pass

        def sibling(self):
            return FakeCtx()

        def close(self):
            pass

    class FakeProb:
        def __init__(self, ctx, X, W2, Y, flags=0):
            self.ctx, self.tag, self.fits = ctx, X, []

        def free(self):
            pass

    def fake_single(prob, rank, alpha_in, rank_tol=.1, rng=None, mode="device", latency_mode=True, **kw):
        calls.append((prob.tag, latency_mode))
        prob.fits = [(0.5, 1, n_iter[prob.tag])]
        return np.array([True]), np.full((1, 1, 1, 1), prob.tag), np.zeros(1), 0.5

    monkeypatch.setattr(capi, "Context", FakeCtx)
    monkeypatch.setattr(pruner, "LayerProblem", FakeProb)
    monkeypatch.setattr(pruner, "prune_layer", fake_single)
    specs = [dict(layer_id=i, N=5000, c=c, n=c, k=3, rank=int(c / 1.15)) for i, c in enumerate([512, 512, 512, 512, 64])]

    def picked():
        return sorted(tag for tag, mode in calls if mode == "gram")

    for adapt, expect in (("1", [[0, 1], [1, 3], [1, 3]]), ("0", [[0, 1], [0, 1], [0, 1]])):
        monkeypatch.setenv("CP_PRECOMPUTE_ADAPT", adapt)
        rset = shard.ResidentLayerSet(0, specs, lambda s: (s["layer_id"], None, None), per_stream=1, precompute_heaviest=2)
        try:
            for want in expect:
                calls.clear()
                out = rset.run()
                assert [int(W[0, 0, 0, 0]) for _, W, _, _ in out] == list(range(5))
                assert picked() == want, (adapt, picked(), want)          # the 64-channel layer (99 epochs) is not in the pool
        finally:
            rset.close()
    n_iter.update({0: 30, 3: 30})                                         # ties: 0, 1, 3 equal -> the later starters (1 and 3)
    monkeypatch.setenv("CP_PRECOMPUTE_ADAPT", "1")
    rset = shard.ResidentLayerSet(0, specs, lambda s: (s["layer_id"], None, None), per_stream=1, precompute_heaviest=2)
    try:
        rset.run()
        calls.clear()
        rset.run()
        assert picked() == [1, 3]
    finally:
        rset.close()


def test_bench_vgg16_job_is_the_reference_rank_table():
    """bench.py's whole-network job: d_c = max(int(c / 1.15), rank) with the reference's rank table x 4/3 (net.py:1309-1327,
    1346-1349) for the 12 conv -> conv pairs; golden names V01..V12 exist for every layer."""
    import bench
    specs = bench.cpjobs.JOBS["vgg16"]()
    assert [(s["c"], s["n"], s["rank"]) for s in specs] == [
        (64, 64, 55), (64, 128, 55), (128, 128, 111), (128, 256, 111), (256, 256, 222), (256, 256, 222), (256, 512, 222),
        (512, 512, 445), (512, 512, 445), (512, 512, 445), (512, 512, 445), (512, 512, 445)]
    for s in specs:
        g = np.load(os.path.join(ROOT, "tests", "golden", s["name"] + ".npz"))
        p = json.loads(str(g["params"]))
        assert (p["layer_id"], p["c"], p["n"], p["rank"], p["N"]) == (s["layer_id"], s["c"], s["n"], s["rank"], s["N"])
        assert "newW2_sketch" in g.files and g["idxs"].shape == (s["c"],)


def test_precompute_flag_modes(monkeypatch):
    """pruner.precompute_flag: single-layer calls (True) ask for the normal equations AND the factorisation of the full Gram
    during the search, a resident set's heaviest layers ("gram") for the normal equations only, batches for nothing; only
    when at least 70 % of the channels are kept; CP_REFIT_PRECOMPUTE=0/1 in the environment forces it."""
    from cpmi355 import capi, pruner
    monkeypatch.delenv("CP_REFIT_PRECOMPUTE", raising=False)
    both = capi.CP_REFIT_PRECOMPUTE | capi.CP_REFIT_PREFACTOR
    assert pruner.precompute_flag(True, 445, 512) == both
    assert pruner.precompute_flag("gram", 445, 512) == capi.CP_REFIT_PRECOMPUTE
    assert pruner.precompute_flag(False, 445, 512) == 0
    assert pruner.precompute_flag(True, 256, 512) == 0 and pruner.precompute_flag(True, 512, 512) == 0
    monkeypatch.setenv("CP_REFIT_PRECOMPUTE", "0")
    assert pruner.precompute_flag(True, 445, 512) == 0
    monkeypatch.setenv("CP_REFIT_PRECOMPUTE", "1")
    assert pruner.precompute_flag("gram", 256, 512) == capi.CP_REFIT_PRECOMPUTE and pruner.precompute_flag(True, 256, 512) == both



def test_job_tables_match_the_goldens():
    """bench.py's synthetic operands (cpmi355.jobs.synth) are the arrays the goldens were generated from"""
    import cp_oracle
    from cpmi355 import jobs
    for spec in (jobs.vgg16_4x()[0], jobs.vgg16_5x()[1], jobs.resnet50_2x()[0], jobs.resnet50_2x()[2]):
        g = np.load(os.path.join(_GOLDEN, spec["name"] + ".npz"))
        p = json.loads(str(g["params"]))
        assert all(p[k] == spec[k] for k in ("layer_id", "N", "c", "n", "k", "rank"))
        small = dict(spec, N=200)
        a = jobs.synth(small)
        b = cp_oracle.synth_layer(p["layer_id"], 200, p["c"], p["n"], p["k"], residual=p.get("residual", False))
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_job_tables_follow_the_reference_fixtures():
    """kept counts of the released models: VGG-16 5x (temp/channel_pruning.prototxt) and ResNet-50 2x
    (temp/resnet-50-cp.prototxt); every layer of the three jobs has a reference golden"""
    from cpmi355 import jobs
    assert [s["rank"] for s in jobs.vgg16_4x()] == [55, 55, 111, 111, 222, 222, 222, 445, 445, 445, 445, 445]
    assert [s["rank"] for s in jobs.vgg16_5x()] == [24, 22, 41, 51, 108, 89, 111, 184, 276, 228]
    assert all(s["N"] == 20000 for s in jobs.vgg16_5x())
    r = jobs.resnet50_2x()
    assert len(r) == 40 and max(s["c"] for s in r) == 2048
    sel = [s for s in r if s["name"].endswith("_sel")]
    assert [s["rank"] for s in sel] == [35, 101, 97, 144, 205, 198, 288, 278, 418, 407, 423, 412, 595, 606, 1222, 1147]
    assert all(s["k"] == 1 and not s["residual"] for s in sel)
    assert all(s["residual"] and s["k"] == 1 and s["n"] == 4 * s["c"] for s in r if s["name"].endswith("_b2b"))
    assert all(s["k"] == 3 and s["n"] == s["c"] for s in r if s["name"].endswith("_b2a"))
    assert len({s["layer_id"] for s in r + jobs.vgg16_5x() + jobs.vgg16_4x()}) == 62
    for s in r + jobs.vgg16_5x() + jobs.vgg16_4x():
        assert os.path.isfile(os.path.join(_GOLDEN, s["name"] + ".npz")), s["name"]


def test_bench_gpus_n_launches_n_ranks(monkeypatch):
    """`python bench.py --gpus N` without a torch.distributed environment re-launches itself under torch.distributed.run
    with N ranks on 127.0.0.1; with one, a WORLD_SIZE that differs from --gpus is refused."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--workload", "resnet50"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--workload", "resnet50"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # inside a 2-rank environment --gpus 4 is an error, not a silent world of 2
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(bench.Env, "__init__", lambda self: setattr(self, "world", 2))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)


def _mask_gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
    from cpmi355 import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = [dict(c=c) for c in (16, 5, 9)]
    res = [(np.random.RandomState(10 * rank + i).rand(s["c"]) < 0.5, None, None) for i, s in enumerate(specs)]
    every = shard.gather_masks(specs, res, dist)
    q.put((rank, [[m.tolist() for m in per_rank] for per_rank in every]))
    dist.destroy_process_group()


def test_gather_masks_world_size_2_gloo():
    """weak scaling's only collective: one uint8 all_gather of the channel masks -> masks[rank][layer] on every rank"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_mask_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1]
    for r in range(2):
        for i, c in enumerate((16, 5, 9)):
            assert got[0][1][r][i] == (np.random.RandomState(10 * r + i).rand(c) < 0.5).tolist()


def test_gemm_launch_covers_every_tile_and_chunk_exactly_once():
    """The f64 GEMM behind the Gram builds divides its 128 x 128 tiles over the workgroups of a launch in three ways (whole
    tiles in an XCD-contiguous super-tile order, the tail tiles split along K, uniform split-K).  cp_debug_gemm_units replays
    the plan and the kernel's own workgroup -> (tile, chunk) mapping on the host: for every shape up to 40 x 40 tiles, every
    triangle mode and several K, each tile of the product is produced exactly once -- by one whole-tile workgroup, or by the
    chunks 0 .. s-1 of a split -- and the chunks cover K."""
    import ctypes
    from cpmi355 import capi
    lib = capi.load()
    plan = (ctypes.c_int32 * 9)()
    seen_modes = set()
    for tri in (0, 1, 2):
        for K in (128, 400, 768, 5008, 20000):
            for tm in list(range(1, 41)):
                for tn in ([tm] if tri else sorted({1, 2, 4, 7, tm, 35, 40})):
                    M, N = tm * 128, tn * 128
                    n = lib.cp_debug_gemm_units(256, M, N, K, tri, plan, None, 0)
                    assert n > 0
                    n_tiles, tiles_n, small, planes, n_full, n_split, s, kchunk, units_n = list(plan)
                    assert units_n == n
                    units = (ctypes.c_int32 * (5 * n))()
                    assert lib.cp_debug_gemm_units(256, M, N, K, tri, plan, units, n) == n
                    u = np.frombuffer(units, dtype=np.int32).reshape(n, 5)
                    live = u[u[:, 4] == 0]
                    edge = 64 if small else 128
                    rows, cols = M // edge, N // edge
                    want = {(i, j) for i in range(rows) for j in range(cols)
                            if tri == 0 or (tri == 1 and j <= i) or (tri == 2 and i <= j)}
                    assert len(want) == n_tiles
                    got = {}
                    for ti, tj, z, nz, _ in live:
                        got.setdefault((int(ti), int(tj)), []).append((int(z), int(nz)))
                    assert set(got) == want, (tri, K, tm, tn)
                    for t, zs in got.items():
                        nz = zs[0][1]
                        assert sorted(z for z, _ in zs) == list(range(nz)) and all(k == nz for _, k in zs), (tri, K, tm, tn, t)
                        if nz > 1:                       # the chunks cover the k range
                            assert kchunk % 16 == 0 and kchunk * nz >= K
                    split_tiles = sum(1 for zs in got.values() if zs[0][1] > 1)
                    if planes:
                        assert split_tiles == n_tiles and n == n_tiles * planes and planes > 8
                        seen_modes.add("planes")
                    elif n_split:
                        assert split_tiles == n_split and 2 <= s <= 8 and n_full + n_split == n_tiles
                        if n_full:                       # tail split: whole rounds of the chip's 512 slots first
                            assert n_full % 512 == 0 and n_split < 512
                            seen_modes.add("tail")
                        else:
                            assert s == 8
                            seen_modes.add("uniform8")
                    else:
                        assert split_tiles == 0 and n == n_tiles
                        seen_modes.add("whole")
    assert seen_modes == {"planes", "tail", "uniform8", "whole"}
