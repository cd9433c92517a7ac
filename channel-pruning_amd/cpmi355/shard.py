"""Multi-GPU sharding of independent layer problems (one process per GPU).

Independent (producer, consumer) conv pairs are embarrassingly parallel once their operands
are frozen: every rank prunes its own subset, no collective touches the
data path; the only communication is the gather of the per-layer results (channel masks, a few
hundred bytes each, plus the reconstructed weights) over torch.distributed -- backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

    assign_layers(costs, world)        longest-processing-time-first assignment
    layer_cost(N, c, n, k, rank)       FLOP model of SURVEY.md section 8d (plus the serial CD term)
    prune_sharded(specs, compute_fn)   run this rank's share, all-gather every result
    GpuLayerBatches(ctx, operands)     compute_many for it: equal-width layers through cp_prune_layers, up to 16 at a time

For the layers that dominate (conv4/conv5 sizes) the ROWS of one layer can be spread over the ranks instead
("secondary"):

    row_range(N, world, rank)          contiguous balanced row slice of a rank
    prune_layer_rows(...)              dictionary() with X / Y row-sharded: the S sampled rows are exchanged once
                                       (a few MB), every rank runs the identical alpha search, the refit's column
                                       sums and Gram are summed with two all-reduces (RCCL), every rank solves
"""
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
        self.flags = (capi.CP_CD_RECIPROCAL | capi.CP_CD_DELTA) if flags is None else flags
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


def prune_sharded(specs, compute_fn=None, dist=None, device=None, compute_many=None):
    """specs: list of dicts with at least N, c, n, k, rank; compute_fn(spec) -> (idxs, W, b), or
    compute_many(list of this rank's specs) -> list of (idxs, W, b) (e.g. a GpuLayerBatches).
    Every rank returns the full list of results in layer order.  `dist` is an initialised
    torch.distributed module (None = single process)."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    costs = [layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"]) for s in specs]
    owner = assign_layers(costs, world)
    mine = {}
    own = [i for i in range(len(specs)) if owner[i] == rank]
    if compute_many is not None:
        got = compute_many([specs[i] for i in own])
    else:
        got = [compute_fn(specs[i]) for i in own]
    for i, (idxs, W, b) in zip(own, got):
        mine[i] = (np.asarray(idxs, dtype=bool), np.asarray(W, dtype=np.float64), np.asarray(b, dtype=np.float64))
    if dist is None:
        return [mine[i] for i in range(len(specs))]
    import torch
    results = [None] * len(specs)
    # masks: one fixed-size uint8 all_gather (the "trivial gather of selected-channel masks")
    cmax = max(s["c"] for s in specs)
    local = torch.zeros((len(specs), cmax), dtype=torch.uint8, device=device)
    for i, (idxs, _, _) in mine.items():
        local[i, : idxs.shape[0]] = torch.from_numpy(idxs.astype(np.uint8)).to(local.device)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    masks = [gathered[owner[i]][i, : specs[i]["c"]].cpu().numpy().astype(bool) for i in range(len(specs))]
    # weights / biases: variable size -> broadcast from the owner
    for i, s in enumerate(specs):
        kept = int(masks[i].sum())
        shape = (s["n"], kept, s["k"], s["k"])
        if owner[i] == rank:
            W = torch.from_numpy(np.ascontiguousarray(mine[i][1].reshape(shape))).to(local.device)
            b = torch.from_numpy(np.ascontiguousarray(mine[i][2])).to(local.device)
        else:
            W = torch.empty(shape, dtype=torch.float64, device=local.device)
            b = torch.empty((s["n"],), dtype=torch.float64, device=local.device)
        dist.broadcast(W, src=owner[i])
        dist.broadcast(b, src=owner[i])
        results[i] = (masks[i], W.cpu().numpy(), b.cpu().numpy())
    return results


# ---------------------------------------------------------------------------------------------------------
# one layer, rows sharded over the ranks
# ---------------------------------------------------------------------------------------------------------
def row_range(N, world, rank):
    """[lo, hi) of `rank` when N rows are split into `world` contiguous, balanced slices."""
    base, extra = divmod(int(N), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_sum(dist, t):
    """In-place sum over the ranks of a torch tensor.  RCCL ("nccl") reduces device tensors in place over xGMI;
    with "gloo" (CPU tests, or several ranks on one GPU) a device tensor is staged through the host."""
    if dist is None or dist.get_world_size() == 1:
        return t
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    elif not t.is_cuda and dist.get_backend() == "nccl":
        import torch
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(d)
        t.copy_(d.cpu())
    else:
        dist.all_reduce(t)
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
        return torch.zeros(int(elems), dtype=torch.float64, device=self.device)

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
