"""Multi-GPU sharding of independent layer problems (one process per GPU).

Independent (producer, consumer) conv pairs are embarrassingly parallel once their operands
are frozen: every rank prunes its own subset, no collective touches the
data path; the only communication is the gather of the per-layer results (channel masks, a few
hundred bytes each, plus the reconstructed weights) over torch.distributed -- backend "nccl"
(= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

    assign_layers(costs, world)        longest-processing-time-first assignment
    layer_cost(N, c, n, k, rank)       FLOP model of SURVEY.md section 8d (plus the serial CD term)
    prune_sharded(specs, compute_fn)   run this rank's share, all-gather every result
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


def prune_sharded(specs, compute_fn, dist=None, device=None):
    """specs: list of dicts with at least N, c, n, k, rank; compute_fn(spec) -> (idxs, W, b).
    Every rank returns the full list of results in layer order.  `dist` is an initialised
    torch.distributed module (None = single process)."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    costs = [layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"]) for s in specs]
    owner = assign_layers(costs, world)
    mine = {}
    for i, s in enumerate(specs):
        if owner[i] == rank:
            idxs, W, b = compute_fn(s)
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
