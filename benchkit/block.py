"""benchkit.block -- BASELINE.json configs[1], the VGG-16 conv3_x block: one instance alone (value_conv3_block of the default
line) and replica throughput (--workload block)."""
import ctypes
import os
import threading
import time

import numpy as np

from .common import (BLOCK_GOLDEN, BLOCK_LAYERS, CD_FLAGS, F64_MFMA_PEAK_TFLOPS, KSIZE, MIN_TIMED_SECONDS, N_SAMPLES, PROFILE_TAG,
                     cpu_model, golden_check, layer_flops, synth)
from .roofline import roofline_object

class LayerWorker(threading.Thread):
    """One layer shape, one HIP stream, one host thread driving it.  batch > 1: the worker holds `batch` independent
    copies of the layer (own operands, own sibling context on the same stream) and prunes them with ONE
    cp_prune_layers call per iteration -- their alpha searches are the workgroups of one launch."""

    def __init__(self, device, layer_id, c, n, rank, batch=1):
        super().__init__(daemon=True)
        import cpmi355
        self.cpmi355 = cpmi355
        self.layer_id, self.c, self.n, self.rank = layer_id, c, n, rank
        self.ctx = cpmi355.Context(device)
        self.ctxs = [self.ctx] + [self.ctx.sibling() for _ in range(batch - 1)]
        X, W2, Y, _ = synth(layer_id, c, n)
        self.X, self.W2, self.Y = X, W2, Y
        self.probs = [cpmi355.LayerProblem(cx, X, W2, Y, flags=CD_FLAGS) for cx in self.ctxs]
        self.prob = self.probs[0]
        self.rngs = [np.random.RandomState(1234 + layer_id) for _ in self.probs]
        self.rng0 = [cpmi355.pruner.rng_mark(r) for r in self.rngs]
        self.go = threading.Event()
        self.done = threading.Event()
        self.stop = False
        self.result = None
        self.error = None
        self.gram_ms, self.gram_flops, self.gram_exec, self.stage_acc, self.host_acc = [], [], [], {}, []
        self.collect = False
        self.single = False          # True: one cp_prune_layer call per pruning even when batch > 1
        self.todo = 1
        self.calls, self.layers_done = 0, 0

    def prune(self, count):
        """`count` (<= batch) independent prunings of this layer; every one starts from the reference's RNG state"""
        rngs = self.rngs[:count]
        for r, mark in zip(rngs, self.rng0):  # = np.random.seed(1234 + id) before every call, without re-seeding
            self.cpmi355.pruner.rng_rewind(r, mark)
        self.calls += 1
        self.layers_done += count
        if count == 1:
            return [self.cpmi355.prune_layer(self.prob, self.rank, 1e-3, rank_tol=.1, rng=rngs[0], mode="device",
                                             latency_mode=self.single)]
        return self.cpmi355.prune_layers_batched(self.probs[:count], [self.rank] * count, [1e-3] * count, rngs, rank_tol=.1)

    def run(self):
        while True:
            self.go.wait()
            self.go.clear()
            if self.stop:
                return
            try:
                left = self.todo
                while left > 0:
                    count = 1 if self.single else min(left, len(self.probs))
                    left -= count
                    self.result = self.prune(count)[0]
                    if self.collect:
                        for prob in self.probs[:count]:
                            for name, ms in prob.ctx.last_stage_times():
                                self.stage_acc.setdefault(name, []).append(ms)
                            p = int(prob.refit_info.p)
                            self.gram_flops.append(float(N_SAMPLES) * p * p)   # symmetric half of 2 N p^2
                            tiles = (p + 127) // 128
                            n_pad = (N_SAMPLES + 15) // 16 * 16       # what the launch executes: lower 128-tiles
                            self.gram_exec.append(tiles * (tiles + 1) // 2 * 128.0 * 128.0 * n_pad * 2.0)
                        if count == 1:
                            ht = (ctypes.c_double * 4)()
                            self.ctx.lib.cp_debug_host_times(ctypes.c_void_p(self.ctx.h), ht)
                            self.host_acc.append(tuple(ht))
            except BaseException as e:  # noqa
                self.error = e
            self.done.set()


def run_passes(groups, passes):
    """`passes` passes over the block.  Pass s is worker group s % D's (own contexts / HIP streams and
    operand copies); every worker runs its share back to back, so up to D x batch independent passes are in flight
    on the GPU (D = 1, batch = 1: strictly one pass at a time)."""
    active = []
    for g, group in enumerate(groups):
        cnt = len(range(g, passes, len(groups)))
        if cnt == 0:
            continue
        for w in group:
            w.todo = cnt
            w.done.clear()
            w.go.set()
            active.append(w)
    for w in active:
        w.done.wait()
        if w.error is not None:
            raise w.error


def block_single_instance(device, passes=7):
    """ONE instance of the conv3_x block: its three (independent) layers side by side on three streams through
    cp_prune_layer, nothing else on the chip.  -> dict (ms per pass, layers/s, per-stage ms, fits, CD steps)"""
    group = [LayerWorker(device, lid, c, n, r, batch=1) for lid, c, n, r in BLOCK_LAYERS]
    for w in group:
        w.start()
        w.single = True
    try:
        for cx in (w.ctx for w in group):
            cx.enable_stage_timing(1)
        ts = []
        WARM = 4      # untimed passes first: this leg follows ~45 s of CPU baseline with an idle GPU, and the first passes after
                      # that ran 2x slower in one run out of three (clocks / runtime state coming back up)
        for i in range(passes + WARM):
            t1 = time.perf_counter()
            for w in group:                      # the three layers of ONE block instance, concurrently (own streams)
                w.collect = i >= WARM
                w.todo = 1
                w.done.clear()
                w.go.set()
            for w in group:
                w.done.wait()
                if w.error is not None:
                    raise w.error
            if i >= WARM:
                ts.append((time.perf_counter() - t1) * 1e3)
        stages = {}
        for w in group:
            for name, v in w.stage_acc.items():
                stages.setdefault(name, []).extend(v)
        g_ms = [ms for w in group for ms in w.stage_acc.get("refit_gram_gemm", [])]
        g_fl = [f for w in group for f in w.gram_flops]
        g_ex = [f for w in group for f in w.gram_exec]
        cd_steps = [sum(f[2] for f in w.prob.fits) * w.c for w in group]
        cd_ms = [float(np.mean(w.stage_acc.get("cd_alpha_search", [0.0]))) for w in group]
        parity, werrs = True, []
        for w in group:
            idxs, newW2, _, _ = w.result
            same, werr = golden_check(BLOCK_GOLDEN[w.layer_id], idxs, newW2)
            werrs.append(werr)
            parity = parity and bool(same) and werr is not None and werr <= 1e-5
        ms = float(np.median(ts))
        return {"ms_per_pass": round(ms, 3), "layers_per_s": round(len(BLOCK_LAYERS) / ms * 1e3, 2),
                "stage_ms_avg": {k: round(sum(v) / len(v), 4) for k, v in stages.items()},
                "lasso_fits_per_layer": [len(w.prob.fits) for w in group], "cd_steps_per_layer": cd_steps,
                "cd_us_per_step": [round(m * 1e3 / max(1, s), 4) for m, s in zip(cd_ms, cd_steps)],
                "mask_parity_vs_reference_golden": parity, "weights_rel_frobenius_vs_reference_golden": werrs,
                "roofline_kernel_alone": ({"achieved": round(sum(g_fl) / (sum(g_ms) * 1e-3) / 1e12, 3),
                                           "frac": round(sum(g_fl) / (sum(g_ms) * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS, 4),
                                           "executed_tflops": round(sum(g_ex) / (sum(g_ms) * 1e-3) / 1e12, 3),
                                           "avg_launch_ms": round(sum(g_ms) / len(g_ms), 4)} if g_ms and sum(g_ms) > 0 else None),
                "host_ms_avg": ({k: round(float(np.mean([h[i] for w in group for h in w.host_acc])), 4) for i, k in
                                 enumerate(("lasso_operands_enqueue", "alpha_search_host", "refit_enqueue",
                                            "copy_back_and_wait"))} if any(w.host_acc for w in group) else None)}, group
    except BaseException:
        close_workers(group)
        raise


def close_workers(workers):
    for w in workers:
        w.stop = True
        w.go.set()
    for w in workers:
        w.join(timeout=10)
    for w in workers:            # release device memory and streams before the interpreter tears modules down
        for prob in w.probs:
            prob.free()
        for cx in reversed(w.ctxs):      # siblings before the context that owns the stream
            cx.close()


def bench_block(args, env):
    rank, world = env.rank, env.world
    depth = max(1, args.inflight)
    batch = max(1, args.batch)
    groups = [[LayerWorker(env.local_rank, lid + 100 * rank if rank else lid, c, n, r, batch=batch)
               for lid, c, n, r in BLOCK_LAYERS] for _ in range(depth)]
    workers = [w for g in groups for w in g]
    for w in workers:
        w.start()
    per_round = depth * batch                      # passes one round of calls covers (every copy once)
    run_passes(groups, per_round)                  # every problem copy runs once
    for w in workers:
        w.ctx.sync()
    t0 = time.perf_counter()
    run_passes(groups, per_round)
    for w in workers:
        w.ctx.sync()
    round_s = time.perf_counter() - t0
    for _ in range(max(0, args.warmup - 2)):
        run_passes(groups, per_round)
    # one step = `rounds` full rounds (depth x batch passes each): K steps take >= MIN_TIMED_SECONDS
    rounds = max(1, int(np.ceil(MIN_TIMED_SECONDS / max(round_s * args.steps, 1e-9))))
    rounds = env.bcast_int(rounds)
    for w in workers:
        for cx in w.ctxs:
            cx.enable_stage_timing(2)    # timed region: only the two events around the roofline kernel
        w.collect = True
        w.calls, w.layers_done = 0, 0
    for w in workers:
        w.ctx.sync()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_passes(groups, per_round * rounds)
    for w in workers:
        w.ctx.sync()
    env.barrier()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    passes = args.steps * per_round * rounds
    out = None
    if rank == 0:
        parity, werrs, recon = True, [], []
        for w in groups[0]:
            idxs, newW2, newB2, _ = w.result
            same, werr = golden_check(BLOCK_GOLDEN.get(w.layer_id, "-"), idxs, newW2)
            werrs.append(werr)
            if same is not None:
                parity = parity and same and werr is not None and werr <= 1e-5
            Xs = w.X[:, idxs].reshape(N_SAMPLES, -1).astype(np.float64)
            res = Xs @ newW2.reshape(w.n, -1).T + newB2 - w.Y
            recon.append(round(float(np.linalg.norm(res) / np.linalg.norm(w.Y)), 6))
        layers_per_s = len(BLOCK_LAYERS) * world * passes / elapsed
        g_ms = [ms for w in workers for ms in w.stage_acc.get("refit_gram_gemm", [])]
        g_fl = [f for w in workers for f in w.gram_flops]
        bcls = {"alpha_search": [ms for w in workers for ms in w.stage_acc.get("cd_alpha_search", [])], "refit_gram": g_ms,
                "cholesky_chain": [], "backward_substitution": [ms for w in workers for ms in w.stage_acc.get("refit_solve", [])]}
        roof = roofline_object(bcls, g_fl, [], passes, workers[0].ctx, PROFILE_TAG, "block")
        fl = [layer_flops(w.c, w.n, int(w.prob.refit_info.p)) for w in groups[0]]
        alg_l = sum(f[0] for f in fl) / len(fl)
        calls = sum(w.calls for w in workers)
        out = {
            "metric": "conv layers pruned/sec (VGG-16 4x, 5k samples)",
            "value": round(layers_per_s, 3), "unit": "layers/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(2, args.warmup), "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "block: REPLICA THROUGHPUT of the VGG-16 conv3_x block 4x prune (3 layers: 128->256, "
                                   "256->256, 256->256; k=3, rank=c/2, N=5000) -- independent copies of the block in "
                                   "flight; see single_instance for one block alone",
                       "passes_timed": passes, "passes_per_step": per_round * rounds, "timed_region_s": round(elapsed, 3),
                       "block_copies_in_flight": depth * batch, "layers_in_flight": 3 * depth * batch, "streams": 3 * depth,
                       "layers_per_call_actual": round(sum(w.layers_done for w in workers) / max(1, calls), 2),
                       "foreign_calls_timed": calls, "parallelism": "replicas x%d" % world},
            "mask_parity_vs_reference_golden": parity, "weights_rel_frobenius_vs_reference_golden": werrs,
            "reconstruction_rel_frobenius_err": recon, "roofline": roof,
            "job_mfma": {"gflop_per_layer_algorithmic": round(alg_l / 1e9, 2),
                         "sustained_tflops_algorithmic_per_gpu": round(layers_per_s / world * alg_l / 1e12, 2),
                         "frac_of_peak_algorithmic": round(layers_per_s / world * alg_l / 1e12 / F64_MFMA_PEAK_TFLOPS, 4)},
        }
    close_workers(workers)
    if rank == 0:
        single, group = block_single_instance(env.local_rank)
        close_workers(group)
        out["single_instance"] = single
        out["single_instance_layers_per_s"] = single["layers_per_s"]
        if world == 1 and not args.no_cpu_baseline:
            bspecs = [dict(layer_id=lid, name="L%02d" % (lid - 30), N=N_SAMPLES, c=c, n=n, k=KSIZE, rank=r) for lid, c, n, r in BLOCK_LAYERS]
            from .cpu_legs import cpu_best_threads, cpu_port_seconds
            best, sweep = cpu_best_threads(bspecs[:1])
            secs = cpu_port_seconds(bspecs, threads=best)
            out["cpu_baseline"] = {"value": round(len(BLOCK_LAYERS) / sum(secs), 4), "unit": "layers/s", "cores": int(best),
                                   "blas_thread_sweep_s": sweep,
                                   "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
                                   "sample": "one pass over the 3 conv3_x layers (N=5000), sklearn Lasso (single-threaded CD) "
                                             "+ LinearRegression/gelsd (BLAS threads = cores, the fastest of the sweep); %.1f s total, per layer %s s" % (
                                                 sum(secs), [round(s, 2) for s in secs]),
                                   "speedup_single_instance_latency": round(sum(secs) * 1e3 / single["ms_per_pass"], 1)}
    return out

