"""Multi-GPU sharding of independent layer problems (one process per GPU).

Independent (producer, consumer) conv pairs are embarrassingly parallel once their operands
are frozen: every rank prunes its own subset, no collective touches the
data path; the only communication is the gather of the per-layer results (channel masks, a few
hundred bytes each, plus the reconstructed weights) over torch.distributed -- backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

    assign_layers(costs, world)        longest-processing-time-first assignment
    layer_cost(N, c, n, k, rank)       FLOP model of SURVEY.md section 8d (plus the serial CD term)
    prune_sharded(specs, compute_fn)   run this rank's share, then exchange_results(): one mask all_gather to every rank +
                                       the owners' packed (W, b) to rank 0 (exchange="gather", the default), to every
                                       rank ("allgather") or nowhere ("masks"); a rank that fails tells the others in the
                                       mask all_gather (ShardPeerError) instead of leaving them waiting
    GpuLayerBatches(ctx, operands)     compute_many for it: equal-width layers through cp_prune_layers, up to 16 at a time
    ResidentLayerSet(device, specs, ..) the same with the operands resident in HBM and every width group on its own
                                       stream(s) + host thread, all in flight together (bench.py --workload vgg16)

For the layers that dominate (conv4/conv5 sizes) the ROWS of one layer can be spread over the ranks instead
("secondary"):

    row_range(N, world, rank)          contiguous balanced row slice of a rank
    prune_layer_rows(...)              dictionary() with X / Y row-sharded: the S sampled rows are exchanged once
                                       (a few MB), every rank runs the identical alpha search, the refit's column
                                       sums and Gram are summed with two all-reduces (RCCL), every rank solves
"""
import os

import numpy as np


def layer_cost(N, c, n, k, rank):
    """Relative cost of one dictionary() call: the section 8d flop count + the sequential CD sweep."""
    kk = k * k
    S = min(400, N // 20)
    p = rank * kk * 1.05
    flops = 2 * c * S * kk * n + 2 * S * n * c * c + 2 * N * p * p + 2 * N * p * n + p ** 3 / 3 + 2 * p * p * n
    cd_steps = 10 * 18 * c                      # ~10 fits x ~18 epochs x c coordinates
    return flops / 30e12 + cd_steps * 150e-9 * max(1.0, c / 256.0)


def assign_layers(costs, world):
    """LPT: heaviest layer first onto the least-loaded rank.  Returns owner[i] for every layer."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += costs[i]
    return owner


class GpuLayerBatches:
    """compute_many for prune_sharded: this rank's layers on the GPU, those of equal channel count `max_batch` at a
    time through cp_prune_layers (their alpha searches side by side in one launch).

        operands(spec) -> (X[N,c,k,k], W2[n,c,k,k], Y[N,n])   host arrays of one layer
        seed(spec)     -> the seed of the layer's own RNG stream (parity is per layer: SURVEY.md section 8e)

    Returns [(idxs, newW2, newB2), ...] in the order of the specs it was given; .alphas holds the accepted alphas."""

    def __init__(self, ctx, operands, seed=lambda s: 1234 + s["layer_id"], max_batch=16, alpha_in=1e-3, rank_tol=.1,
                 flags=None):
        from . import capi
        self.ctx, self.operands, self.seed = ctx, operands, seed
        self.max_batch, self.alpha_in, self.rank_tol = int(max_batch), alpha_in, rank_tol
        self.flags = 0 if flags is None else flags      # 0 = sklearn's operation order (capi.CP_CD_* trade it for speed)
        self.alphas = {}

    def __call__(self, specs):
        from .pruner import LayerProblem, prune_layers_batched
        out = [None] * len(specs)
        by_width = {}
        for i, s in enumerate(specs):
            by_width.setdefault(int(s["c"]), []).append(i)
        siblings = [self.ctx.sibling() for _ in range(min(self.max_batch, max(map(len, by_width.values()), default=1)) - 1)]
        ctxs = [self.ctx] + siblings
        try:
            for c, members in sorted(by_width.items()):
                for g0 in range(0, len(members), self.max_batch):
                    group = members[g0:g0 + self.max_batch]
                    probs = []
                    try:
                        for cx, i in zip(ctxs, group):
                            X, W2, Y = self.operands(specs[i])
                            probs.append(LayerProblem(cx, X, W2, Y, flags=self.flags))
                        rngs = [np.random.RandomState(self.seed(specs[i])) for i in group]
                        res = prune_layers_batched(probs, [specs[i]["rank"] for i in group],
                                                   [specs[i].get("alpha_in", self.alpha_in) for i in group], rngs,
                                                   rank_tol=self.rank_tol)
                        for i, (idxs, W, b, alpha) in zip(group, res):
                            out[i] = (idxs, W, b)
                            self.alphas[specs[i].get("layer_id", i)] = alpha
                    finally:
                        for pr in probs:
                            pr.free()
        finally:
            for cx in siblings:
                cx.close()
        return out


class ResidentLayerSet:
    """This rank's layers RESIDENT in HBM, pruned as one job.

    The layers are grouped by channel count (cp_prune_layers wants equal c) and every group is cut into chunks of at
    most `per_stream` layers; a chunk owns a Context (= one HIP stream / hardware queue) plus sibling contexts and a
    host thread, so all chunks -- all widths -- are in flight together: the alpha searches of a chunk are the
    workgroups of one launch, the refits of different chunks overlap.  Operands are uploaded once (constructor);
    run() prunes every layer once and returns [(idxs, newW2, newB2, alpha), ...] in the order of `specs`; parity is
    per layer (own RandomState(seed(spec)), fixed alpha_in: SURVEY.md section 8e).  borrow_results: newW2 / newB2 are
    views of each layer's page-locked result block (Context.result_host) instead of fresh arrays -- no host copy, and
    exchange_results DMAs straight from them; they are valid until the next run() / close().

        operands(spec) -> (X[N,c,k,k], W2[n,c,k,k], Y[N,n]) host arrays, called once per layer."""

    def __init__(self, device, specs, operands, seed=lambda s: 1234 + s["layer_id"], per_stream=2, alpha_in=1e-3,
                 rank_tol=.1, flags=None, precompute_heaviest=None, borrow_results=False):
        import threading

        from . import capi
        from .pruner import LayerProblem, rng_mark
        self.specs = list(specs)
        self.alpha_in, self.rank_tol = alpha_in, rank_tol
        flags = 0 if flags is None else flags           # 0 = sklearn's operation order (capi.CP_CD_* trade it for speed)
        by_width = {}
        for i, s in enumerate(self.specs):
            by_width.setdefault(int(s["c"]), []).append(i)
        self.chunks = []                 # dicts: members, ctxs, probs, rngs, marks, thread plumbing
        for c, members in sorted(by_width.items(), key=lambda kv: -kv[0]):     # widest (slowest) first
            # per_stream: layers per chunk -- one number, or {channel count: number} (default 1 for the counts not named)
            per = per_stream.get(c, per_stream.get("default", 1)) if isinstance(per_stream, dict) else per_stream
            per = max(1, min(int(per), capi_max_jobs()))
            for g0 in range(0, len(members), per):
                group = members[g0:g0 + per]
                # (stream priorities were measured both ways -- the widest layers' streams higher: vgg16 job 29.5 against
                #  28.0 ms; the narrow layers' lower: no change -- and are not used)
                root = capi.Context(device)
                ctxs = [root] + [root.sibling() for _ in group[1:]]
                probs, rngs = [], []
                for cx, i in zip(ctxs, group):
                    X, W2, Y = operands(self.specs[i])
                    probs.append(LayerProblem(cx, X, W2, Y, flags=flags))
                    probs[-1].borrow_results = bool(borrow_results)
                    rngs.append(np.random.RandomState(seed(self.specs[i])))
                self.chunks.append(dict(members=group, ctxs=ctxs, probs=probs, rngs=rngs,
                                        marks=[rng_mark(r) for r in rngs], go=threading.Event(), done=threading.Event(),
                                        out=None, error=None, ms=0.0))
        # The heaviest layers' alpha searches are the idle head of the job (one workgroup each for milliseconds): that many
        # of them get their full normal equations computed on the side stream meanwhile (pruner.precompute_flag); more
        # than the idle head can absorb only adds flops to a chip that is busy afterwards.
        if precompute_heaviest is None:
            precompute_heaviest = 2      # vgg16 job: 29.7 / 28.0 / 27.9 / 28.0 / 28.9 / 30.0 ms with 0 .. 5
        cost_of = lambda ch: layer_cost(*[self.specs[ch["members"][0]][k] for k in ("N", "c", "n", "k", "rank")])   # noqa: E731
        single = [ch for ch in self.chunks if len(ch["members"]) == 1]
        single.sort(key=lambda ch: -cost_of(ch))
        self._latency_chunks = set(id(ch) for ch in single[:max(0, precompute_heaviest)]) if len(self.chunks) > 2 else \
            set(id(ch) for ch in single)
        # WHICH of the equally heavy layers get it is re-decided after every run from what the run showed (finish()): the ones
        # whose alpha searches were the longest -- their refits start last, and the precompute takes Gram and X^T Y off exactly
        # that path.  The cost model cannot know: the number of fits of a search depends on the data (6, 7 or 8 for the five
        # 512-channel layers of the vgg16 job).  vgg16 job 22.7-22.9 against 23.5-23.7 ms with the first two of the five
        # (three processes each, alternating; CP_PRECOMPUTE_ADAPT=0 keeps the first choice).  Masks do not depend on the
        # choice; a layer's coefficients move at the 1e-16 level when it changes sides (another summation order).
        import os
        self._adapt = len(self.chunks) > 2 and precompute_heaviest > 0 and os.environ.get("CP_PRECOMPUTE_ADAPT", "1") != "0"
        top = cost_of(single[0]) if single else 0.0
        self._adapt_pool = [ch for ch in single if cost_of(ch) >= 0.9 * top]       # the heaviest class (equal widths)
        self._adapt_n = min(max(0, precompute_heaviest), len(self._adapt_pool))
        # a set of one or two layers has the chip to itself: the full treatment (pruner.precompute_flag)
        self._latency_kind = "gram" if len(self.chunks) > 2 else True
        self._stop = False
        self._threads = []
        self._t_go = 0.0
        for ch, nxt in zip(self.chunks, self.chunks[1:] + [None]):
            ch["next"] = nxt
        for ch in self.chunks:
            t = threading.Thread(target=self._worker, args=(ch,), daemon=True)
            t.start()
            self._threads.append(t)

    def _prune_chunk(self, ch):
        import time

        from .pruner import prune_layer, prune_layers_batched, rng_rewind
        ch["lag_ms"] = (time.perf_counter() - self._t_go) * 1e3     # from start() to this chunk's thread getting going
        for r, m in zip(ch["rngs"], ch["marks"]):          # every run starts from the layer's own seed
            rng_rewind(r, m)
        specs = [self.specs[i] for i in ch["members"]]
        t0 = time.perf_counter()
        if len(specs) == 1:
            s = specs[0]
            out = [prune_layer(ch["probs"][0], s["rank"], s.get("alpha_in", self.alpha_in), rank_tol=self.rank_tol,
                               rng=ch["rngs"][0], mode="device",
                               latency_mode=self._latency_kind if id(ch) in self._latency_chunks else False)]
        else:
            out = prune_layers_batched(ch["probs"], [s["rank"] for s in specs],
                                       [s.get("alpha_in", self.alpha_in) for s in specs], ch["rngs"],
                                       rank_tol=self.rank_tol)
        ch["ms"] = (time.perf_counter() - t0) * 1e3
        ch["end_ms"] = (time.perf_counter() - self._t_go) * 1e3
        return out

    def _worker(self, ch):
        while True:
            ch["go"].wait()
            ch["go"].clear()
            if self._stop:
                return
            # The chunks start as a CHAIN, widest layers first: a thread hands the baton to the next chunk before its own
            # host-side preamble (RNG draws, argument marshalling: ~70 us under the interpreter lock), so the next thread gets
            # the lock the moment this one enters its foreign call.  With all twelve woken at once the order in which they got
            # the lock was arbitrary and a 512-channel layer -- the job's critical path -- started up to 0.9 ms late.
            nxt = ch.get("next")
            if nxt is not None and not self._stop:
                nxt["go"].set()
            try:
                ch["out"] = self._prune_chunk(ch)
            except BaseException as e:   # noqa
                ch["error"] = e
            ch["done"].set()

    def start(self):
        """every chunk begins its layers now; collect with wait() (some of the layers, as they finish) and / or finish()"""
        import time
        self._t_go = time.perf_counter()
        for ch in self.chunks:
            ch["done"].clear()
            ch["error"] = None
        if self.chunks:
            self.chunks[0]["go"].set()          # the others follow as a chain (see _worker)

    def wait(self, indices):
        """blocks until the layers `indices` (positions in the constructor's specs) are done -> {index: (idxs, W, b, alpha)}.
        The chunks of the other layers keep running."""
        want = set(int(i) for i in indices)
        out = {}
        for ch in self.chunks:
            if want.isdisjoint(ch["members"]):
                continue
            ch["done"].wait()
            if ch["error"] is not None:
                self.finish(raise_errors=False)      # nothing may still be inside its foreign call when this propagates
                raise ch["error"]
            for i, r in zip(ch["members"], ch["out"]):
                if i in want:
                    out[i] = r
        return out

    def finish(self, raise_errors=True):
        """every chunk is done -> all results in layer order"""
        out = [None] * len(self.specs)
        for ch in self.chunks:          # every chunk finishes before anything is raised: a chunk still inside its foreign
            ch["done"].wait()           # call must not see its events re-armed or its contexts freed
        for ch in self.chunks:
            if ch["error"] is not None:
                if raise_errors:
                    raise ch["error"]
                continue
            for i, r in zip(ch["members"], ch["out"]):
                out[i] = r
        if self._adapt and all(ch["error"] is None for ch in self.chunks):
            # coordinate steps of a layer's search = sum of n_iter over its fits x channels (LayerProblem.fits); the later
            # starter wins a tie (its search ends later)
            order = {id(ch): k for k, ch in enumerate(self.chunks)}
            steps = lambda ch: sum(f[2] for f in (ch["probs"][0].fits or [])) * int(self.specs[ch["members"][0]]["c"])   # noqa: E731
            ranked = sorted(self._adapt_pool, key=lambda ch: (-steps(ch), -order[id(ch)]))
            others = self._latency_chunks - set(id(ch) for ch in self._adapt_pool)
            self._latency_chunks = others | set(id(ch) for ch in ranked[:self._adapt_n])
        return out

    def run(self):
        self.start()
        return self.finish()

    def __call__(self, specs=None):
        """compute_many of prune_sharded (the specs are the ones given to the constructor)."""
        return [(idxs, W, b) for idxs, W, b, _ in self.run()]

    def chunk_report(self):
        return [dict(layers=[self.specs[i].get("name", self.specs[i].get("layer_id", i)) for i in ch["members"]],
                     c=int(self.specs[ch["members"][0]]["c"]), ms=round(ch["ms"], 3), start_lag_ms=round(ch.get("lag_ms", 0.0), 3),
                     end_ms=round(ch.get("end_ms", 0.0), 3)) for ch in self.chunks]

    def problems(self):
        """{index in specs: LayerProblem} (fit logs, refit_info of the last run)"""
        return {i: pr for ch in self.chunks for i, pr in zip(ch["members"], ch["probs"])}

    def close(self):
        self._stop = True
        for ch in self.chunks:
            ch["go"].set()
        for t in self._threads:
            t.join(timeout=60)
        for ch, t in zip(self.chunks, self._threads):
            if t.is_alive():            # still inside a foreign call: leave its device memory and contexts alone (leaked, not
                continue                # freed under a running kernel)
            for pr in ch["probs"]:
                pr.free()
            for cx in reversed(ch["ctxs"]):
                cx.close()
        self.chunks = []


def capi_max_jobs():
    return 16      # CP_MAX_JOBS (include/cpmi355.h)


def _all_gather_rows(dist, local):
    """(world,) + local.shape, on local's device.  "nccl" (= RCCL): ONE all_gather_into_tensor on device memory over
    xGMI -- all seven links of a GPU carry traffic at once, unlike a ring broadcast per owner; any other backend
    (gloo: CPU tests, several ranks on one GPU) is staged through host tensors."""
    import torch
    world = dist.get_world_size()
    if dist.get_backend() == "nccl":
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    host = local.cpu().contiguous()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host)
    return torch.stack(parts).to(local.device)


def result_segments(specs, owner, masks, world):
    """Packed layout of the exchange: owner r's segment = its layers in layer order, each as W [n, kept, k, k] then b [n]
    (float64 elements).  -> (shapes per layer, offset of every layer inside its owner's segment, segment length per rank)."""
    shapes = [(int(s["n"]), int(np.count_nonzero(m)), int(s["k"]), int(s["k"])) for s, m in zip(specs, masks)]
    seg = [0] * world
    offs = [0] * len(specs)
    for i, s in enumerate(specs):
        offs[i] = seg[owner[i]]
        seg[owner[i]] += int(np.prod(shapes[i])) + int(s["n"])
    return shapes, offs, seg


LAST_EXCHANGE_MS = {}      # wall time of the phases of this process's last exchange_results (host clock)


def gather_masks(specs, results, dist, device=None):
    """The "trivial gather of selected-channel masks" alone: ONE fixed-size uint8 all_gather of every rank's masks of the
    layers in `specs` (results[i][0] = idxs of layer i on this rank) -> masks[rank][layer] (bool).  Used when every rank
    prunes its own instance of the job (weak scaling): the weights stay with the instance that produced them."""
    import torch
    cmax = max(s["c"] for s in specs)
    local = np.zeros((len(specs), cmax), dtype=np.uint8)
    for i, r in enumerate(results):
        local[i, : specs[i]["c"]] = np.asarray(r[0], dtype=np.uint8)
    on_gpu = dist.get_backend() == "nccl"
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else torch.device("cpu")
    g = _all_gather_rows(dist, torch.from_numpy(local).to(dev)).cpu().numpy()
    return [[g[r, i, : specs[i]["c"]].astype(bool) for i in range(len(specs))] for r in range(g.shape[0])]


class ShardPeerError(RuntimeError):
    """a rank of the job failed before the exchange; raised on EVERY rank by exchange_results (ranks: who)"""

    def __init__(self, ranks):
        super().__init__("cpmi355.shard: rank(s) %s failed before the exchange of the results" % ", ".join(map(str, ranks)))
        self.ranks = list(ranks)


RANK_FAILED = 0xFF       # every byte of a rank's row block in the mask all_gather: "my layers are not coming" (a mask holds 0 / 1)


XGMI_LINK_GBPS = 153.0      # one xGMI link of an MI355X (the GPUs of a node are fully connected: one link per pair)
PCIE_GBPS = 50.0            # device -> page-locked host memory


def exchange_model_ms(seg_bytes, mode, root=0):
    """What the exchange of the packed (W, b) segments costs on an 8-GPU MI355X node, modelled (no such node was available to
    measure on): every pair of GPUs has its own xGMI link, so both a gather to `root` and a full all_gather are bounded by
    the LARGEST segment crossing ONE link (all links run at once); what differs is the copy back to host memory -- the root
    alone (gather) or every rank (allgather) pulls the others' segments over PCIe.  -> dict(xgmi_ms, d2h_ms)."""
    seg_bytes = [float(b) for b in seg_bytes]
    if mode == "masks" or len(seg_bytes) < 2:
        return dict(xgmi_ms=0.0, d2h_ms=0.0)
    others = [b for r, b in enumerate(seg_bytes) if r != root] if mode == "gather" else seg_bytes
    xgmi = max(others) / (XGMI_LINK_GBPS * 1e9) * 1e3 + 0.02
    recv = (sum(seg_bytes) - seg_bytes[root]) if mode == "gather" else (sum(seg_bytes) - min(seg_bytes))
    return dict(xgmi_ms=round(xgmi, 3), d2h_ms=round(recv / (PCIE_GBPS * 1e9) * 1e3, 3))


def exchange_results(specs, owner, mine, dist, device=None, staging=None, failed=False, mode="gather", root=0):
    """north_star's split of the exchange: (1) ONE fixed-size uint8 all_gather of the channel masks -- every rank ends with
    every layer's mask, and the sizes of everything else follow from them, so nothing has to be negotiated; (2) the owners'
    packed float64 results (result_segments), by `mode`:
        "gather" (default)   to rank `root` only, every owner's segment at its exact length (point-to-point sends, one per
                             owner: a gather of uneven parts), so that `root` ends with every layer's (mask, W, b) and the
                             other ranks with (mask, None, None) for the layers they do not own;
        "allgather"          ONE all_gather padded to the longest segment: every rank ends with everything;
        "masks"              nothing travels but the masks (the weights stay with their owner).
    staging "device" (default with backend "nccl" = RCCL over xGMI): this rank's (W, b) go host -> HBM once (a DMA straight
    out of the page-locked result blocks when the ResidentLayerSet lends them: borrow_results), the collective runs between
    the GPUs, and what arrives comes back through ONE page-locked host buffer whose slices are the returned arrays.
    staging "host" (default otherwise): NumPy buffers through the backend.
    failed: this rank could not prune its layers.  It still joins collective (1) -- with its block filled with RANK_FAILED --
    so that nobody is left waiting in it; every rank then raises ShardPeerError before step (2)."""
    import time

    import torch
    if mode not in ("gather", "allgather", "masks"):
        raise ValueError("exchange_results: mode %r" % (mode,))
    t_begin = time.perf_counter()
    world, rank = dist.get_world_size(), dist.get_rank()
    if staging is None:
        staging = "device" if dist.get_backend() == "nccl" else "host"
    on_gpu = staging == "device"
    dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if on_gpu else torch.device("cpu")
    cmax = max(s["c"] for s in specs)
    local = np.zeros((len(specs), cmax), dtype=np.uint8)
    if failed:
        local[:] = RANK_FAILED
    else:
        for i, (idxs, _, _) in mine.items():
            local[i, : idxs.shape[0]] = idxs
    gathered = _all_gather_rows(dist, torch.from_numpy(local).to(dev)).cpu().numpy()
    bad = [r for r in range(world) if gathered[r].size and int(gathered[r].max()) == RANK_FAILED]
    if bad:
        raise ShardPeerError(bad)
    masks = [gathered[owner[i], i, : specs[i]["c"]].astype(bool) for i in range(len(specs))]
    shapes, offs, seg = result_segments(specs, owner, masks, world)
    maxseg = max(1, max(seg))
    t_masks = time.perf_counter()
    t_gather = t_masks
    receives = (mode == "allgather" and world > 1) or (mode == "gather" and rank == root)
    sends = mode == "allgather" or (mode == "gather" and rank != root and seg[rank] > 0)
    starts, flat = {}, None                      # starts[r]: where rank r's segment begins in `flat`
    if mode != "masks" and (sends or receives):
        if on_gpu:
            send = torch.empty(maxseg if mode == "allgather" else max(1, seg[rank]), dtype=torch.float64, device=dev)
        else:
            send = np.zeros(maxseg if mode == "allgather" else max(1, seg[rank]), dtype=np.float64)
        if sends:
            for i, (_, W, b) in mine.items():
                nw = int(np.prod(shapes[i]))
                if on_gpu:
                    send[offs[i]:offs[i] + nw].copy_(torch.from_numpy(np.ascontiguousarray(W).reshape(-1)), non_blocking=True)
                    send[offs[i] + nw:offs[i] + nw + specs[i]["n"]].copy_(torch.from_numpy(np.ascontiguousarray(b)),
                                                                         non_blocking=True)
                else:
                    send[offs[i]:offs[i] + nw] = np.asarray(W).reshape(-1)
                    send[offs[i] + nw:offs[i] + nw + specs[i]["n"]] = b
        if mode == "allgather":
            every = _all_gather_rows(dist, send if on_gpu else torch.from_numpy(send))
            parts = {r: every[r, :seg[r]] for r in range(world) if r != rank and seg[r]}
        else:
            # a gather of uneven parts: the root posts one receive per owner, every other owner one send.  Device tensors only
            # through "nccl" (= RCCL: grouped ncclSend / ncclRecv over xGMI); any other backend gets host tensors, as in
            # _all_gather_rows (gloo handed CUDA tensors for send / recv took 2 s per 50 MB: two ranks on one GPU, round 6)
            ops, parts = [], {}
            p2p_dev = dev if (on_gpu and dist.get_backend() == "nccl") else torch.device("cpu")
            if rank == root:
                for r in range(world):
                    if r != root and seg[r]:
                        parts[r] = torch.empty(seg[r], dtype=torch.float64, device=p2p_dev)
                        ops.append(dist.P2POp(dist.irecv, parts[r], r))
            elif sends:
                out_t = send if on_gpu else torch.from_numpy(send)
                ops.append(dist.P2POp(dist.isend, out_t if out_t.device == p2p_dev else out_t.to(p2p_dev), root))
            for req in (dist.batch_isend_irecv(ops) if ops else []):
                req.wait()
        t_gather = time.perf_counter()
        total = 0
        for r in sorted(parts):
            starts[r] = total
            total += seg[r]
        if on_gpu:
            back = torch.empty(max(1, total), dtype=torch.float64, pin_memory=True)
            for r, st in starts.items():
                back[st:st + seg[r]].copy_(parts[r], non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            flat = back.numpy()
        else:
            flat = np.empty(max(1, total), dtype=np.float64)
            for r, st in starts.items():
                flat[st:st + seg[r]] = parts[r].numpy() if hasattr(parts[r], "numpy") else parts[r]
    results = [None] * len(specs)
    for i, s in enumerate(specs):
        if owner[i] == rank:
            results[i] = (masks[i], np.asarray(mine[i][1]).reshape(shapes[i]), np.asarray(mine[i][2]))
        elif owner[i] in starts:
            o = starts[owner[i]] + offs[i]
            nw = int(np.prod(shapes[i]))
            results[i] = (masks[i], flat[o:o + nw].reshape(shapes[i]), flat[o + nw:o + nw + s["n"]])
        else:
            results[i] = (masks[i], None, None)      # "gather" on a rank that is not the root, or "masks"
    t_end = time.perf_counter()
    sent = int(seg[rank]) * 8 if sends else 0                    # this rank's segment (what it contributes)
    link_out = sent * ((world - 1) if mode == "allgather" else 1)    # ... and what leaves it over its xGMI links
    model = exchange_model_ms([v * 8 for v in seg], mode, root)
    LAST_EXCHANGE_MS.update(total=(t_end - t_begin) * 1e3, masks=(t_masks - t_begin) * 1e3,
                            pack_and_gather=(t_gather - t_masks) * 1e3, back_to_host=(t_end - t_gather) * 1e3, mode=mode,
                            bytes_sent=sent, link_bytes_out=link_out, bytes_received=int(sum(seg[r] for r in starts)) * 8,
                            mask_bytes_sent=int(local.size), padded_segment_bytes=int(maxseg) * 8,
                            xgmi_model_ms=model["xgmi_ms"], d2h_model_ms=model["d2h_ms"])
    return results


def prune_sharded(specs, compute_fn=None, dist=None, device=None, compute_many=None, owner=None, staging=None,
                  force_exchange=False, rounds=None, exchange="gather"):
    """specs: list of dicts with at least N, c, n, k, rank; compute_fn(spec) -> (idxs, W, b), or
    compute_many(list of this rank's specs) -> list of (idxs, W, b) (a GpuLayerBatches or a ResidentLayerSet).
    Every rank returns the list of results in layer order -- every layer's mask everywhere; the weights where `exchange` puts
    them ("gather", the default: all of them on rank 0, (mask, None, None) elsewhere for foreign layers; "allgather":
    everything everywhere; "masks": only the owner has them; see exchange_results).  `dist` is an initialised
    torch.distributed module (None = single process); owner[i] (default: LPT over layer_cost) says which rank
    prunes layer i.  force_exchange: run the two collectives even in a process group of ONE rank (the RCCL path on a
    one-GPU box: all_gather_into_tensor on device tensors, the page-locked return buffer)."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if owner is None:
        owner = plan_owners(specs, world)
    # rounds (plan_rounds): the exchange of the early layers overlaps the pruning of the heavy ones -- needs a compute_many
    # with start() / wait() / finish() (ResidentLayerSet, ThreadedLayerSet)
    if rounds is not None and len(set(rounds)) > 1 and dist is not None and (world > 1 or force_exchange) and \
            compute_many is not None and hasattr(compute_many, "wait"):
        return prune_sharded_rounds(specs, compute_many, dist, owner, rounds, device, staging, exchange)
    mine = {}
    own = [i for i in range(len(specs)) if owner[i] == rank]
    exchanging = dist is not None and (world > 1 or force_exchange)
    try:
        if compute_many is not None:
            got = compute_many([specs[i] for i in own])
        else:
            got = [compute_fn(specs[i]) for i in own]
        for i, (idxs, W, b) in zip(own, got):
            mine[i] = (np.asarray(idxs, dtype=bool), np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64))
    except BaseException as e:   # noqa
        if exchanging:           # the other ranks are on their way into the mask all_gather: tell them there, then raise here
            try:
                exchange_results(specs, owner, {}, dist, device, staging, failed=True, mode=exchange)
            except ShardPeerError:
                pass
        raise e
    if not exchanging:
        return [mine[i] for i in range(len(specs))]
    return exchange_results(specs, owner, mine, dist, device, staging, mode=exchange)


def plan_rounds(specs, owner=None, heavy_fraction=0.5):
    """Exchange round of every layer for prune_sharded(..., rounds=...): 0 for the layers that finish early, 1 for the heavy
    ones (cost above heavy_fraction of the job's most expensive layer).  The results of round 0 travel -- one mask all_gather
    + one all_gather of the packed (W, b), as always -- while the heavy layers are still being pruned; the job then ends with
    the exchange of the heavy layers alone.  Every rank computes the same table (costs are part of the specs)."""
    costs = [s.get("cost", layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"])) for s in specs]
    top = max(costs) if costs else 0.0
    rounds = [1 if c > heavy_fraction * top else 0 for c in costs]
    if len(set(rounds)) < 2:
        rounds = [0] * len(specs)
    return rounds


class ThreadedLayerSet:
    """The start() / wait() / finish() protocol of ResidentLayerSet around a plain compute_fn(spec) -> (idxs, W, b), one
    host thread per layer: what prune_sharded(rounds=...) needs from its compute_many.  (CPU tests; any engine that prunes
    a layer per call.)"""

    def __init__(self, specs, compute_fn):
        self.specs, self.compute_fn = list(specs), compute_fn
        self._threads, self._out, self._err = [], {}, {}

    def start(self):
        import threading
        self._out, self._err = {}, {}

        def work(i):
            try:
                self._out[i] = self.compute_fn(self.specs[i])
            except BaseException as e:   # noqa
                self._err[i] = e

        self._threads = [threading.Thread(target=work, args=(i,), daemon=True) for i in range(len(self.specs))]
        for t in self._threads:
            t.start()

    def wait(self, indices):
        out = {}
        for i in indices:
            self._threads[i].join()
            if i in self._err:
                self.finish(raise_errors=False)
                raise self._err[i]
            out[i] = self._out[i] + (None,) if len(self._out[i]) == 3 else self._out[i]
        return out

    def finish(self, raise_errors=True):
        for t in self._threads:
            t.join()
        if raise_errors and self._err:
            raise next(iter(self._err.values()))
        return [self._out.get(i) for i in range(len(self.specs))]

    def __call__(self, specs=None):
        self.start()
        return [r[:3] for r in self.finish()]


def prune_sharded_rounds(specs, layer_set, dist, owner, rounds, device=None, staging=None, exchange="gather"):
    """prune_sharded with the exchange OVERLAPPED with the pruning: `layer_set` (a ResidentLayerSet / ThreadedLayerSet over
    THIS rank's layers, in layer order) is started, and for every round r = 0, 1, ... this rank waits for ITS layers of
    that round only and joins the exchange of that round's layers (exchange_results on the sub-list: the same two
    collectives, the same packed lay-out) while its later rounds are still running on the GPU.  Every rank walks the
    rounds in the same order, so the collectives match.  -> every layer's (mask, W, b) in layer order, on every rank.
    LAST_EXCHANGE_MS: the last round's figures + "rounds": [ms per round]."""
    import time
    world, rank = dist.get_world_size(), dist.get_rank()
    own = [i for i in range(len(specs)) if owner[i] == rank]
    pos = {i: k for k, i in enumerate(own)}              # layer index -> position in layer_set
    layer_set.start()
    results = [None] * len(specs)
    per_round, totals = [], {}
    try:
        for r in sorted(set(rounds)):
            members = [i for i in range(len(specs)) if rounds[i] == r]
            try:
                got = layer_set.wait([pos[i] for i in members if i in pos])
            except BaseException as e:   # noqa   (as prune_sharded: the other ranks learn it in this round's mask all_gather)
                try:
                    exchange_results([specs[i] for i in members], [owner[i] for i in members], {}, dist, device, staging,
                                     failed=True, mode=exchange)
                except ShardPeerError:
                    pass
                raise e
            mine = {}
            for k, i in enumerate(members):
                if i in pos:
                    idxs, W, b = got[pos[i]][:3]
                    mine[k] = (np.asarray(idxs, dtype=bool), np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64))
            t0 = time.perf_counter()
            sub = exchange_results([specs[i] for i in members], [owner[i] for i in members], mine, dist, device, staging,
                                   mode=exchange)
            per_round.append((time.perf_counter() - t0) * 1e3)
            for k_ in ("bytes_sent", "link_bytes_out", "bytes_received", "xgmi_model_ms", "d2h_model_ms"):
                totals[k_] = totals.get(k_, 0) + LAST_EXCHANGE_MS.get(k_, 0)
            for k, i in enumerate(members):
                results[i] = sub[k]
    finally:
        layer_set.finish(raise_errors=False)
    LAST_EXCHANGE_MS["rounds"] = [round(v, 3) for v in per_round]
    LAST_EXCHANGE_MS["total"] = float(sum(per_round))
    LAST_EXCHANGE_MS.update(totals)             # bytes and modelled times: the sum over the rounds
    return results


def plan_owners(specs, world):
    """owner[i] of every layer: LPT over the cost model (a spec may carry a measured "cost" that overrides it)."""
    costs = [s.get("cost", layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"])) for s in specs]
    return assign_layers(costs, world)


# ---------------------------------------------------------------------------------------------------------
# one layer, rows sharded over the ranks
# ---------------------------------------------------------------------------------------------------------
def row_shard_cost_test(spec, ranks=2, gemm_tflops=50.0, link_gb_s=150.0):
    """Does spreading the ROWS of this layer over `ranks` GPUs (prune_layer_rows) pay?  -> dict(saved_ms, cost_ms, pays).
    What it divides is the refit's Gram and X^T Y (N p^2 + 2 N p n flop at the rate the GEMM sustains in a job); what it
    costs is the two all-reduces of the normal equations (8 p^2 + 8 p n bytes each way over one xGMI link per pair of ranks:
    a ring all-reduce moves 2 (r - 1) / r of the buffer) plus one of the column sums; the alpha search -- the larger part of
    a layer -- is not divided.  p = kept channels x k^2 (5 % slack on the requested rank, as layer_cost)."""
    kk = spec["k"] ** 2
    p = spec["rank"] * kk * 1.05
    N, n = spec["N"], spec["n"]
    flops = N * p * p + 2.0 * N * p * n
    saved_ms = flops * (1.0 - 1.0 / ranks) / (gemm_tflops * 1e12) * 1e3
    bytes_ar = (8.0 * p * p + 8.0 * p * n) * 2.0 * (ranks - 1) / ranks
    cost_ms = bytes_ar / (link_gb_s * 1e9) * 1e3 + 0.05       # + the launch / rendezvous of three collectives
    return dict(saved_ms=round(saved_ms, 3), cost_ms=round(cost_ms, 3), pays=bool(saved_ms > 1.5 * cost_ms))


def row_range(N, world, rank):
    """[lo, hi) of `rank` when N rows are split into `world` contiguous, balanced slices."""
    base, extra = divmod(int(N), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _collective_device(device=None):
    """The GPU a host tensor is staged on for an RCCL collective.  Callers on worker threads MUST name it: torch's current
    device is per thread (a fresh thread starts on device 0 whatever the main thread set), so falling back to
    torch.cuda.current_device() from AssistedJob's threads would put rank r's tensor on cuda:0 -- a duplicate-GPU error or a
    hang inside the communicator of cuda:r."""
    import torch
    if device is None:
        return torch.device("cuda", torch.cuda.current_device())
    return device if isinstance(device, torch.device) else torch.device("cuda", int(device))


def allreduce_sum(dist, t, group=None, device=None):
    """In-place sum over the ranks (of `group`, default: all) of a torch tensor.  RCCL ("nccl") reduces device tensors in
    place over xGMI; with "gloo" (CPU tests, or several ranks on one GPU) a device tensor is staged through the host.
    device: where a HOST tensor is staged with RCCL (default: the calling thread's current device -- name it on worker
    threads, see _collective_device)."""
    if dist is None or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    elif not t.is_cuda and dist.get_backend() == "nccl":
        d = t.to(_collective_device(device))
        dist.all_reduce(d, group=group)
        t.copy_(d.cpu())
    else:
        dist.all_reduce(t, group=group)
    if t.is_cuda:
        import torch
        torch.cuda.synchronize(t.device)   # the library runs on its own stream
    return t


class RowShardEngine:
    """The per-rank arithmetic of prune_layer_rows on the GPU (libcpmi355 through capi.Context).
    tests/test_host_logic.py swaps in a NumPy stand-in with the same methods to exercise the exchange on CPU."""

    def __init__(self, ctx, flags=0):
        import torch
        self.ctx, self.flags = ctx, flags
        self.device = torch.device("cuda", ctx.device)
        self.fits = []
        try:
            torch.cuda.init()
        except RuntimeError as e:   # the system HIP runtime got into the process first (see capi.load)
            raise RuntimeError("torch.cuda cannot initialise after libcpmi355 loaded the system HIP runtime: import torch "
                               "(or set CP_PRELOAD_TORCH=1) before creating the first cpmi355 Context") from e

    def buffer(self, elems):
        import torch
        t = torch.zeros(int(elems), dtype=torch.float64, device=self.device)
        torch.cuda.synchronize(self.device)   # the fill runs on torch's stream, the library writes on its own (non-blocking) one
        return t

    def select(self, Xs, W2, Ys, rank, alpha_in, rank_tol, rng):
        """alpha search on the S exchanged rows (identical on every rank) -> (idxs, alpha)"""
        from .pruner import LayerProblem
        prob = LayerProblem(self.ctx, Xs, W2, Ys, flags=self.flags)
        try:
            prob.lasso_gram(np.arange(Xs.shape[0], dtype=np.int64))
            alpha = prob.alpha_search(rank, alpha_in, rank_tol, rng, mode="device")
            self.fits = list(prob.fits)
            return prob.mask(), alpha
        finally:
            prob.free()

    def load_rows(self, X_local, Y_local):
        from .pruner import _np_dtype_code
        X_local = np.ascontiguousarray(X_local)
        if X_local.dtype not in (np.float32, np.float64):
            X_local = X_local.astype(np.float64)
        self.x_dtype = _np_dtype_code(X_local)
        self.N_local, self.c = int(X_local.shape[0]), int(X_local.shape[1])
        self.kk = int(np.prod(X_local.shape[2:]))
        self.n = int(Y_local.shape[1])
        self.Xd = self.ctx.to_device(X_local)
        self.Yd = self.ctx.to_device(np.ascontiguousarray(Y_local, dtype=np.float64))

    def layout(self, kept):
        return self.ctx.refit_shard_layout(kept, self.kk, self.n)

    def sums(self, mask, sums):
        self.ctx.refit_shard_sums(self.Xd, self.x_dtype, self.N_local, self.c, self.kk, mask, self.Yd, self.n, sums)

    def gram(self, mask, N_total, sums, gram):
        self.ctx.refit_shard_gram(self.Xd, self.x_dtype, self.N_local, self.c, self.kk, mask, self.Yd, self.n, N_total,
                                  sums, gram)

    def solve(self, kept, N_total, ridge, sums, gram):
        p = kept * self.kk
        Wd, bd = self.ctx.empty(self.n * p * 8), self.ctx.empty(self.n * 8)
        try:
            self.refit_info = self.ctx.refit_shard_solve(kept, self.kk, self.n, N_total, ridge, sums, gram, Wd, bd)
            return self.ctx.to_host(Wd, (self.n, p), np.float64), self.ctx.to_host(bd, (self.n,), np.float64)
        finally:
            Wd.free()
            bd.free()

    def free(self):
        for name in ("Xd", "Yd"):
            buf = getattr(self, name, None)
            if buf is not None:
                buf.free()
                setattr(self, name, None)


def prune_layer_rows(engine, X_local, W2, Y_local, row0, N_total, rank, alpha_in, dist=None, rank_tol=.1, rng=None,
                     ridge=0.0, alpha_arg=1e-4, timings=None):
    """dictionary() (lib/decompose.py:386-634) on a layer whose N_total rows are spread over the ranks; this rank
    holds rows [row0, row0 + len(X_local)).  Every rank must enter with the same RNG state (the reference's draws
    -- the sample subset, one seed per fit -- are then identical everywhere) and gets the same
    (idxs, newW2[n, nnz, k, k], newB2, alpha_out) back.

    Exchanges: the S = min(400, N_total // 20) sampled rows of X and Y (each owned by exactly one rank; a sum
    all-reduce of zero-filled buffers is exact), then the two all-reduces of the refit (column sums, Gram).
    An engine whose rows are already resident (engine.load_rows called by the caller) is used as it is and not
    freed.  timings: optional dict, filled with seconds per phase."""
    import time

    import torch
    rng = np.random if rng is None else rng
    X_local = np.asarray(X_local)
    Y_local = np.asarray(Y_local, dtype=np.float64)
    W2 = np.asarray(W2)
    N_local, c = X_local.shape[0], X_local.shape[1]
    k = X_local.shape[2] if X_local.ndim > 2 else 1
    n = W2.shape[0]
    t_last = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
            t_last[0] = now

    samples = rng.randint(0, N_total, min(400, N_total // 20))               # decompose.py:425
    if rank == c:                                                             # decompose.py:487-488
        idxs, alpha = np.array([True] * rank), alpha_arg
    else:
        own = (samples >= row0) & (samples < row0 + N_local)
        Xs = np.zeros((samples.shape[0],) + X_local.shape[1:], dtype=X_local.dtype)
        Ys = np.zeros((samples.shape[0], n), dtype=np.float64)
        Xs[own] = X_local[samples[own] - row0]
        Ys[own] = Y_local[samples[own] - row0]
        Xs = allreduce_sum(dist, torch.from_numpy(Xs)).numpy()
        Ys = allreduce_sum(dist, torch.from_numpy(Ys)).numpy()
        lap("exchange_sampled_rows")
        idxs, alpha = engine.select(Xs, W2, Ys, rank, alpha_in, rank_tol, rng)
        lap("alpha_search")
    kept = int(idxs.sum())
    mask = idxs.astype(np.uint8)
    own_rows = getattr(engine, "Xd", None) is None
    if own_rows:
        engine.load_rows(X_local, Y_local)
        lap("upload_rows")
    try:
        sums_elems, gram_elems = engine.layout(kept)
        sums, gram = engine.buffer(sums_elems), engine.buffer(gram_elems)
        engine.sums(mask, sums)
        lap("refit_sums")
        allreduce_sum(dist, sums)
        lap("allreduce_sums")
        engine.gram(mask, N_total, sums, gram)
        lap("refit_gram")
        allreduce_sum(dist, gram)
        lap("allreduce_gram")
        W, b = engine.solve(kept, N_total, ridge, sums, gram)
        lap("refit_solve")
    finally:
        if own_rows:
            engine.free()
    return idxs, W.reshape((n, kept, k, k)), b, alpha


# ---------------------------------------------------------------------------------------------------------
# one layer whose OWNER is helped by a rank with slack: the rows of the refit split over the two
# ---------------------------------------------------------------------------------------------------------
def plan_assists(specs, owner, world, min_gain_ms=0.3):
    """{layer index: helper rank} for a job sharded by `owner`: the layers for which splitting the refit's rows over two ranks
    pays (row_shard_cost_test) get, heaviest first, the least-loaded rank that (a) is not the owner, (b) helps nobody else
    and (c) has its own layers done well before the owner's alpha search ends (own load <= half the layer's cost) -- the
    helper joins when the owner broadcasts the mask, so it must be idle by then.  Every rank computes the same plan."""
    costs = [s.get("cost", layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"])) for s in specs]
    load = [0.0] * world
    for c_, o in zip(costs, owner):
        load[o] += c_
    assists, taken = {}, set()
    cand = sorted(range(len(specs)), key=lambda i: -costs[i])
    for i in cand:
        t = row_shard_cost_test(specs[i])
        if not t["pays"] or t["saved_ms"] - t["cost_ms"] < min_gain_ms:
            continue
        free = [r for r in range(world) if r != owner[i] and r not in taken and load[r] <= 0.5 * costs[i]]
        if not free:
            continue
        h = min(free, key=lambda r: (load[r], r))
        assists[i] = h
        taken.add(h)
    return assists


OWNER_FAILED = 0xFF      # every byte of the broadcast mask: "the owner's search failed, nothing follows" (a mask holds 0 / 1)


class AssistPeerError(RuntimeError):
    """the OTHER rank of an (owner, helper) pair reported a failure; this rank's own work was fine"""


def _bcast_mask(dist, mask_u8, src, group, device=None):
    """mask uint8[c] from global rank `src` to the ranks of `group` (a device tensor with RCCL -- on `device`, see
    _collective_device --, a host tensor otherwise)"""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(mask_u8, dtype=np.uint8))
    if dist.get_backend() == "nccl":
        t = t.to(_collective_device(device))
    dist.broadcast(t, src=src, group=group)
    return t.cpu().numpy()


def prune_layer_assisted(engine, role, dist, group, owner_rank, c, W2, N_total, rank, alpha_in, X=None, Y=None, rank_tol=.1,
                         rng=None, ridge=0.0, alpha_arg=1e-4, timings=None, before_collectives=None):
    """dictionary() (lib/decompose.py:386-634) on a layer whose owner is HELPED by a second rank for the refit.

    role "owner": X / Y are the layer's host arrays (all N_total rows: the sample subset of decompose.py:425 is drawn from
    them), the engine holds the owner's share of the rows (engine.load_rows, any contiguous part); it runs the alpha search
    alone -- the one step no second GPU can divide -- broadcasts the mask, and both ranks then sum their rows' column sums
    and normal equations (two all-reduces inside `group`); the owner solves.  -> (idxs, newW2[n, nnz, k, k], newB2, alpha).
    role "helper": the engine holds the other rows; waits for the mask, contributes its sums and Gram, returns None.  It
    draws nothing from any RNG.  The reference's RNG stream on the owner is consumed exactly as dictionary() does.

    Failure on either side never leaves the other one blocked in a collective: the sequence broadcast -> all-reduce ->
    all-reduce is walked by BOTH ranks whatever happens.  An owner whose search raises broadcasts the sentinel mask
    (every byte OWNER_FAILED) and re-raises; the helper raises AssistPeerError on it; nothing follows.  After a good mask,
    a rank whose sums / Gram stage raises keeps joining the two all-reduces with whatever its buffers hold and raises its
    own error afterwards; the last element of the Gram all-reduce is a failure count, so the other rank raises
    AssistPeerError instead of solving with half the rows.
    before_collectives: optional callable run after the (collective-free) alpha search and before the first collective --
    AssistedJob's ticket that serialises the collective phases of a rank's assisted layers in layer order."""
    import time

    import torch
    t_last = [time.perf_counter()]
    device = getattr(engine, "device", None)
    if device is not None and getattr(device, "type", "cpu") == "cuda":
        torch.cuda.set_device(device)        # worker threads start on device 0: collectives and staging follow the engine

    def lap(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + now - t_last[0]
            t_last[0] = now

    failure = None
    if role == "owner":
        idxs = alpha = None
        try:
            rng = np.random if rng is None else rng
            X = np.asarray(X)
            Y = np.asarray(Y, dtype=np.float64)
            k = X.shape[2] if X.ndim > 2 else 1
            n = W2.shape[0]
            samples = rng.randint(0, N_total, min(400, N_total // 20))               # decompose.py:425
            if rank == c:                                                             # decompose.py:487-488
                idxs, alpha = np.array([True] * rank), alpha_arg
            else:
                idxs, alpha = engine.select(np.ascontiguousarray(X[samples]), W2, np.ascontiguousarray(Y[samples]), rank,
                                            alpha_in, rank_tol, rng)
            mask = idxs.astype(np.uint8)
        except BaseException as e:   # noqa
            failure, mask = e, np.full(c, OWNER_FAILED, dtype=np.uint8)
        lap("alpha_search")
        if before_collectives is not None:
            before_collectives()
        _bcast_mask(dist, mask, owner_rank, group, device)
        if failure is not None:
            raise failure
    else:
        if before_collectives is not None:
            before_collectives()
        mask = _bcast_mask(dist, np.zeros(c, dtype=np.uint8), owner_rank, group, device)
        if mask.size and int(mask.max()) == OWNER_FAILED:
            raise AssistPeerError("prune_layer_assisted: the owner (rank %d) failed before it had a channel mask" % owner_rank)
    lap("mask_broadcast")
    kept = int(mask.sum())
    sums_elems, gram_elems = engine.layout(kept)       # pure arithmetic on (kept, k^2, n): the same on both ranks, or raises on both
    sums = gram_all = None
    try:
        sums = engine.buffer(sums_elems)
        gram_all = engine.buffer(gram_elems + 1)           # + the failure count of the pair
        engine.sums(mask, sums)
    except BaseException as e:   # noqa
        failure = e
        # not even the buffers: join the collectives with host stand-ins of the same length
        sums = sums if sums is not None else torch.zeros(int(sums_elems), dtype=torch.float64)
        gram_all = gram_all if gram_all is not None else torch.zeros(int(gram_elems) + 1, dtype=torch.float64)
    gram = gram_all[:gram_elems]
    allreduce_sum(dist, sums, group, device)
    lap("refit_sums")
    if failure is None:
        try:
            engine.gram(mask, N_total, sums, gram)
        except BaseException as e:   # noqa
            failure = e
    if failure is not None:
        gram_all[-1] = 1.0
    allreduce_sum(dist, gram_all, group, device)
    lap("refit_gram")
    if failure is not None:
        raise failure
    if float(gram_all[-1]) != 0.0:
        raise AssistPeerError("prune_layer_assisted: the other rank of the pair failed in its sums / Gram stage")
    if role != "owner":
        return None
    W, b = engine.solve(kept, N_total, ridge, sums, gram)
    lap("refit_solve")
    return idxs, W.reshape((n, kept, k, k)), b, alpha


class AssistedJob:
    """compute_many for prune_sharded when some layers of a job are row-assisted (plan_assists): this rank's other layers
    run on its ResidentLayerSet as always; every assisted layer it OWNS and every layer it HELPS runs prune_layer_assisted
    on a host thread of its own with a RowShardEngine (own context = own stream) -- the helper thread blocks on the owner's
    mask broadcast, i.e. it starts working when the owner's search ends, by which time the helper's own (light) layers are
    done.  One process group per (owner, helper) pair, created by every rank in the same order.

        specs, owner, assists   the job, its owner table, {layer index: helper rank}
        operands(spec)          host arrays (X, W2, Y) of a layer, as for ResidentLayerSet
        make_engine()           -> a fresh RowShardEngine (the GPU one, or a stand-in with the same methods)
    """

    def __init__(self, specs, owner, assists, dist, operands, make_engine, rset, rset_index, seed=lambda s: 1234 + s["layer_id"],
                 alpha_in=1e-3, rank_tol=.1):
        self.specs, self.owner, self.assists, self.dist = specs, owner, dict(assists), dist
        self.rset, self.rset_index = rset, list(rset_index)           # rset over specs[i] for i in rset_index (this rank's other layers)
        self.alpha_in, self.rank_tol, self.seed = alpha_in, rank_tol, seed
        rank = dist.get_rank()
        self.groups, self.mine, self.helped = {}, {}, {}
        for i in sorted(self.assists):                                # every rank, same order: collective
            self.groups[i] = dist.new_group(ranks=sorted({owner[i], self.assists[i]}))
        for i, h in sorted(self.assists.items()):
            s = specs[i]
            cut = s["N"] // 2
            if owner[i] == rank:
                X, W2, Y = operands(s)
                eng = make_engine()
                eng.load_rows(X[:cut], Y[:cut])
                self.mine[i] = dict(engine=eng, X=X, W2=W2, Y=Y)
            elif h == rank:
                X, W2, Y = operands(s)
                eng = make_engine()
                eng.load_rows(X[cut:], Y[cut:])
                self.helped[i] = dict(engine=eng, W2=W2)
        self.last_timings = {}

    def __call__(self, own_specs=None):
        import threading
        rank = self.dist.get_rank()
        out, errors = {}, []
        # One thread per assisted layer this rank owns or helps: the alpha searches (no collective) run side by side, but the
        # collective phases of a rank are walked ONE AT A TIME in ascending layer order (tickets) -- two threads of a process
        # inside two communicators at once is not something RCCL promises to survive, and with every rank following the same
        # total order the pair of the lowest unfinished layer is always free to proceed.
        order = sorted(list(self.mine) + list(self.helped))
        done = {i: threading.Event() for i in order}

        def ticket(i):
            before = [j for j in order if j < i]
            return lambda: [done[j].wait() for j in before]

        def own_layer(i):
            try:
                s, m = self.specs[i], self.mine[i]
                tm = {}
                out[i] = prune_layer_assisted(m["engine"], "owner", self.dist, self.groups[i], rank, s["c"], m["W2"], s["N"], s["rank"],
                                              s.get("alpha_in", self.alpha_in), X=m["X"], Y=m["Y"], rank_tol=self.rank_tol,
                                              rng=np.random.RandomState(self.seed(s)), timings=tm,
                                              before_collectives=ticket(i))[:3]
                self.last_timings[s.get("name", i)] = {k: round(v * 1e3, 3) for k, v in tm.items()}
            except BaseException as e:   # noqa
                errors.append(e)
            finally:
                done[i].set()

        def help_layer(i):
            try:
                s, m = self.specs[i], self.helped[i]
                prune_layer_assisted(m["engine"], "helper", self.dist, self.groups[i], self.owner[i], s["c"], m["W2"], s["N"], s["rank"],
                                     s.get("alpha_in", self.alpha_in), before_collectives=ticket(i))
            except BaseException as e:   # noqa
                errors.append(e)
            finally:
                done[i].set()

        threads = [threading.Thread(target=own_layer, args=(i,)) for i in self.mine] + \
                  [threading.Thread(target=help_layer, args=(i,)) for i in self.helped]
        for t in threads:
            t.start()
        rest_error = None
        try:
            rest = self.rset() if self.rset is not None else []
        except BaseException as e:   # noqa   (the assisted threads still have peers waiting for them: join before raising)
            rest, rest_error = [], e
        for t in threads:
            t.join()
        if rest_error is not None:
            raise rest_error
        if errors:
            # this rank's own failure first; an AssistPeerError only says that the OTHER rank of a pair has one
            own_errors = [e for e in errors if not isinstance(e, AssistPeerError)]
            raise (own_errors or errors)[0]
        for i, r in zip(self.rset_index, rest):
            out[i] = r
        return [out[i] for i in sorted(out)]           # this rank's layers in layer order, as prune_sharded expects

    def close(self):
        for m in list(self.mine.values()) + list(self.helped.values()):
            m["engine"].free()
            cx = getattr(m["engine"], "owned_ctx", None)
            if cx is not None:
                cx.close()
