"""benchkit.sequential -- `--sequential-alpha`: the vgg16 job in the reference's own order, cfgs.alpha carried."""
import os
import time

import numpy as np

from .common import CD_FLAGS, JOB_TEXT, ROOT, cpjobs

def bench_sequential_alpha(args, env):
    """`--sequential-alpha`: the 12 layers of the vgg16 job one after another on ONE GPU with the reference's alpha carry --
    what Net.R3's loop does (/root/reference/lib/net.py:1407-1457 calls dictionary() layer by layer and cfgs.alpha, written
    at decompose.py:626-627, is the next call's right bracket, :491).  Nothing overlaps: layer l + 1 needs layer l's alpha.
    Checked against the UNMODIFIED reference run the same way (tests/golden/C01_vgg16_alpha_chain.npz, oracle/gen_golden.py
    --chain): masks and the chain of carried alphas identical; the CPU port run the same way is the cpu_baseline of the line."""
    import cpmi355
    from cpmi355.pruner import LayerProblem, prune_layer
    specs = cpjobs.JOBS["vgg16"]()
    ctx = cpmi355.Context(env.local_rank)
    probs = []
    for spec in specs:
        X, W2, Y, _ = cpjobs.synth(spec)
        probs.append(LayerProblem(ctx, X, W2, Y, flags=CD_FLAGS))

    def one_pass():
        alpha, res = 1e-3, []
        for spec, pr in zip(specs, probs):
            idxs, W, b, alpha = prune_layer(pr, spec["rank"], alpha, rank_tol=.1, rng=np.random.RandomState(1234 + spec["layer_id"]),
                                            mode="device")
            res.append((idxs, alpha))
        return res

    for _ in range(max(1, args.warmup)):
        res = one_pass()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = one_pass()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    job_ms = elapsed / args.steps * 1e3
    out = {"metric": JOB_TEXT["vgg16"][1] + ", sequential alpha carry", "value": round(len(specs) * args.steps / elapsed, 3),
           "unit": "layers/s", "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": round(job_ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "vgg16 --sequential-alpha: the 12 conv->conv pairs one after another, every alpha search starting from "
                                  "the previous layer's final alpha (the reference's cfgs.alpha carry); 1 step = 1 pass over the 12 layers",
                      "layers_per_job": len(specs)},
           "job_ms": round(job_ms, 3), "alpha_chain": [float(a) for _, a in res]}
    chain_path = os.path.join(ROOT, "tests", "golden", "C01_vgg16_alpha_chain.npz")
    if os.path.exists(chain_path):
        g = np.load(chain_path)
        out["masks_and_alpha_chain_identical_to_the_reference_chain"] = bool(
            all(np.array_equal(res[i][0], g["idxs_%02d" % i]) and res[i][1] == float(g["alpha_out"][i]) for i in range(len(specs))))
        out["reference_chain_seconds"] = round(float(np.sum(g["ref_seconds"])), 1)
    if not args.no_cpu_baseline:
        from .cpu_legs import cpu_baseline_object
        cpu_masks = []
        out["cpu_baseline"] = cpu_baseline_object(specs, specs, {}, job_ms, True, carry_alpha=True, masks_out=cpu_masks)
        out["masks_identical_to_cpu_port_with_carry"] = bool(all(np.array_equal(g[0], c_[0]) and g[1] == c_[1]
                                                                  for g, c_ in zip(res, cpu_masks)))
    for pr in probs:
        pr.free()
    ctx.close()
    return out

