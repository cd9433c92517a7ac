"""benchkit.job -- the whole-network jobs (vgg16 = north_star's job, resnet50, vgg16_5x): the timed region and what is
measured around it (layers alone, PCIe-inclusive pass, two jobs in flight, replica throughput, parity against the goldens)."""
import ctypes
import os
import sys
import time

import numpy as np

from .common import (CD_FLAGS, F64_MFMA_PEAK_TFLOPS, JOB_TEXT, MIN_TIMED_SECONDS, PROFILE_TAG, VGG16_COST_MS, algorithmic_bytes,
                     cpjobs, golden_check, layer_flops)
from .roofline import roofline_object

def bench_job(args, env, job):
    import cpmi355
    from cpmi355 import shard
    from cpmi355.pruner import prune_layer, rng_rewind

    specs = cpjobs.JOBS[job]()
    for s in specs:       # measured single-layer latencies (ms, profiles/r02_*) as LPT costs; model when absent
        s["cost"] = (VGG16_COST_MS.get(s["c"], None) if job == "vgg16" else None) or \
            shard.layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"])
    # --scaling strong (default): the layers of ONE job instance are sharded over the ranks (LPT) and every rank ends with
    # every layer's (mask, W, b) (one mask all_gather + one all_gather of the packed results); weak: every rank prunes its OWN
    # instance of the whole job (per-GPU work fixed; the only collective is the uint8 all_gather of the channel masks)
    weak = args.scaling == "weak"
    owner = [env.rank] * len(specs) if weak else shard.plan_owners(specs, env.world)
    own = [i for i in range(len(specs)) if owner[i] == env.rank]
    # N > 1, strong: the layers for which splitting the refit's rows over two ranks pays (shard.row_shard_cost_test: the
    # N = 20000 job's wide layers, never the 5000-sample jobs) get a helper among the ranks with slack (shard.plan_assists):
    # the owner searches and solves, the helper contributes half of the column sums and of the normal equations
    assists = {}
    if not weak and env.dist is not None and not args.no_row_assist:
        assists = shard.plan_assists(specs, owner, env.world)
        if os.environ.get("CP_BENCH_ASSISTS"):      # flow tests on a small box: "layer index:helper rank,..." instead of the plan
            assists = {int(a.split(":")[0]): int(a.split(":")[1]) for a in os.environ["CP_BENCH_ASSISTS"].split(",")}
    rset_index = [i for i in own if i not in assists]
    host_data = {}

    def operands(spec):
        if spec["layer_id"] not in host_data:
            X, W2, Y, _ = cpjobs.synth(spec)
            host_data[spec["layer_id"]] = (X, W2, Y)
        return host_data[spec["layer_id"]]

    if env.world == 1 and not args.profile_mode:
        for sp_ in specs:
            operands(sp_)
    # ---- PCIe-inclusive, BEFORE the resident set exists (the state of a process that just calls dictionary()): every layer
    # ---- pruned from its pageable host arrays, one after another ----
    pcie = None
    if env.world == 1 and not args.profile_mode:
        from cpmi355.pruner import LayerProblem
        ctx0 = cpmi355.Context(env.local_rank)     # a context of its own, as a caller of dictionary() has (default_context)

        seq_layer_ms = {}

        def sequential_pass(x_dtype):
            t1 = time.perf_counter()
            h2d = 0
            for spec in specs:
                X, W2, Y = host_data[spec["layer_id"]]
                t_l = time.perf_counter()
                pr = LayerProblem(ctx0, X.astype(x_dtype, copy=False), W2, Y, flags=CD_FLAGS, defer_upload=True)   # as dictionary() does
                h2d += pr.h2d_bytes
                prune_layer(pr, spec["rank"], 1e-3, rank_tol=.1, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
                pr.free()
                seq_layer_ms[spec["name"]] = round((time.perf_counter() - t_l) * 1e3, 2)
            return time.perf_counter() - t1, h2d

        first, _ = sequential_pass(np.float32)     # first pass: workspaces of the context grow layer by layer (cold)
        t_seq, h2d = sequential_pass(np.float32)   # steady state
        pcie = {"job_ms_sequential_with_h2d": round(t_seq * 1e3, 2), "first_pass_ms": round(first * 1e3, 2),
                "h2d_bytes": int(h2d), "layers_per_s_with_h2d": round(len(specs) / t_seq, 2), "per_layer_ms": dict(seq_layer_ms),
                "note": "every layer pruned from pageable host arrays, one after another: what the drop-in dictionary() does per "
                        "call (cp_prune_layer_h2d: the sampled rows first, X and Y streamed in behind the alpha search); never part of `value`.  first_pass_ms: the "
                        "context's workspaces still growing from layer to layer.  X as float32 (the bytes the reference's "
                        "float64 arrays hold: Caffe blobs); x_float64 = the same pass with X uploaded as the float64 array "
                        "the reference hands to dictionary() (2x the bytes, the astype() excluded)"}
        if not args.no_pcie_f64:
            x64 = {lid: v[0].astype(np.float64) for lid, v in host_data.items()}
            saved = dict(host_data)
            for lid in x64:
                host_data[lid] = (x64[lid],) + saved[lid][1:]
            t64, h64 = sequential_pass(np.float64)
            host_data.update(saved)
            del x64
            pcie["x_float64"] = {"job_ms_sequential_with_h2d": round(t64 * 1e3, 2), "h2d_bytes": int(h64),
                                 "layers_per_s_with_h2d": round(len(specs) / t64, 2)}

        ctx0.close()

    t_up0 = time.perf_counter()
    per_stream = args.per_stream or (1 if job != "resnet50" else 2)
    if job == "resnet50" and not args.per_stream:
        # the two 2048-channel selections own the critical path (their alpha searches, ~25 ms each): a stream each, so that
        # neither waits for the other's Gram and refit (34.9 against 37.5 ms per job); the other widths two layers per stream
        per_stream = {"default": 2, 2048: 1}
    if os.environ.get("CP_BENCH_PER_STREAM_BY_WIDTH"):      # e.g. "512:5,256:3": layers per chunk by channel count
        per_stream = dict(per_stream) if isinstance(per_stream, dict) else {"default": per_stream}
        for item in os.environ["CP_BENCH_PER_STREAM_BY_WIDTH"].split(","):
            k_, v_ = item.split(":")
            per_stream[int(k_)] = int(v_)
    rset = shard.ResidentLayerSet(env.local_rank, [specs[i] for i in rset_index], operands, per_stream=per_stream,
                                  flags=CD_FLAGS, borrow_results=True, precompute_heaviest=args.precompute_heaviest)
    probs = rset.problems()           # index in `rset_index` order -> LayerProblem
    job_set = rset
    if assists:
        def make_engine():
            cx_ = cpmi355.Context(env.local_rank)
            eng_ = shard.RowShardEngine(cx_, flags=CD_FLAGS)
            eng_.owned_ctx = cx_
            return eng_
        job_set = shard.AssistedJob(specs, owner, assists, env.dist, operands, make_engine, rset, rset_index)
    ctxs = [cx for ch in rset.chunks for cx in ch["ctxs"]]
    roots = [ch["ctxs"][0] for ch in rset.chunks]

    masks_equal = [True]
    exchange_rounds = None if (weak or args.no_exchange_rounds) else shard.plan_rounds(specs, owner)

    def one_job():
        if weak:
            res = rset()
            if env.dist is not None:          # the trivial gather of the selected-channel masks (one uint8 all_gather)
                t_x = time.perf_counter()
                every = shard.gather_masks(specs, res, env.dist)
                masks_equal[0] = masks_equal[0] and all(np.array_equal(every[r][i], res[i][0])
                                                        for r in range(env.world) for i in range(len(specs)))
                shard.LAST_EXCHANGE_MS.clear()
                shard.LAST_EXCHANGE_MS.update(total=(time.perf_counter() - t_x) * 1e3, bytes_sent=0, bytes_received=0, mode="masks",
                                              mask_bytes_sent=sum(s["c"] for s in specs), xgmi_model_ms=0.0, d2h_model_ms=0.0)
            return res
        # N > 1: the results of the light layers are exchanged while the heavy ones are still being pruned (shard.plan_rounds)
        return shard.prune_sharded(specs, compute_many=job_set, dist=env.dist, owner=owner, exchange=args.exchange,
                                   staging="device" if env.dist is not None else None,
                                   rounds=exchange_rounds if (env.dist is not None and not assists) else None)

    def sync_all():
        for cx in roots:
            cx.sync()

    def xty_flops(pr):
        """2 N p n when the layer's last refit formed X^T Y in the Gram's launch (its bracket then spans both), else 0"""
        fn = getattr(pr.ctx.lib, "cp_debug_last_xty_fused", None)
        if fn is None:
            return 0.0
        fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
        return 2.0 * float(pr.N) * int(pr.refit_info.p) * float(pr.n) if fn(pr.ctx.h) == 1 else 0.0

    # ---- warm-up: 1 + W jobs, then choose jobs_per_step so that K steps take >= MIN_TIMED_SECONDS ----
    one_job()
    sync_all()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(max(1, args.warmup)):
        one_job()
    sync_all()
    env.barrier()
    job_s = env.max_over_ranks((time.perf_counter() - t0) / max(1, args.warmup))
    reps = args.jobs_per_step or max(1, int(np.ceil(MIN_TIMED_SECONDS / max(job_s * args.steps, 1e-9))))
    reps = env.bcast_int(reps)

    # The stage brackets (HIP events on the launch streams, cp_enable_stage_timing mode 2) are taken during the TIMED jobs, on
    # every STAGE_SAMPLE-th of them: reading them back costs ~1 ms of host time per job (12 layers x ~8 brackets x two
    # hipEventElapsedTime each), which is measurement, not pruning work -- with every job instrumented it was 4 % of `value`
    STAGE_SAMPLE = 8
    g_ms, g_fl, exch_ms = [], [], []
    g_fl_bracket, g_n = [], []         # flops the Gram bracket's launch executed (+ X^T Y when fused), n per bracket
    cls_ms = {"alpha_search": [], "refit_gram": [], "cholesky_chain": [], "backward_substitution": []}   # per launch / bracket, in the job
    chol_fl, chol_steps, chol_pn = [], [], []
    sync_all()
    env.barrier()
    t0 = time.perf_counter()
    epoch0 = time.time()
    windows = {"refit_gram": [], "cholesky_chain": []}     # per job: wall window the concurrent brackets of a class span (ms)
    cd_steps_ns = {}                                        # channel count -> [ns per coordinate step, in the job]
    job_no = 0
    for _ in range(args.steps):
        for _ in range(reps):
            sampled = job_no % STAGE_SAMPLE == 0
            job_no += 1
            if sampled:
                for cx in ctxs:
                    cx.enable_stage_timing(2)      # only the events around the roofline kernels and the two chains
                if roots:
                    roots[0].stage_epoch()       # one clock for the brackets of all the layers' streams (cp_last_stage_spans)
            results = one_job()
            if env.dist is not None:
                exch_ms.append(shard.LAST_EXCHANGE_MS.get("total", 0.0))
            if not sampled:
                continue
            span = {"refit_gram": [], "cholesky_chain": []}
            for j, pr in probs.items():
                for name, ms, begin in pr.ctx.last_stage_spans(roots[0]):
                    if name == "refit_gram_gemm":
                        g_ms.append(ms)
                        # (the launch also forms X^T Y when the library fused the two products: cp_gemm_gram_xty)
                        g_fl.append(float(pr.N) * int(pr.refit_info.p) ** 2)
                        g_fl_bracket.append(g_fl[-1] + xty_flops(pr))
                        g_n.append(float(pr.n))
                        cls_ms["refit_gram"].append(ms)
                        span["refit_gram"].append((begin, begin + ms))
                    elif name == "cd_alpha_search":
                        cls_ms["alpha_search"].append(ms)
                        steps_ = sum(f[2] for f in pr.fits) * pr.c
                        if steps_ > 0:
                            cd_steps_ns.setdefault(pr.c, []).append(ms * 1e6 / steps_)
                    elif name == "refit_cholesky":
                        cls_ms["cholesky_chain"].append(ms)
                        pp = float(int(pr.refit_info.p))
                        chol_fl.append(pp ** 3 / 3.0 + pp * pp * float(pr.n))    # + the forward substitution riding along
                        chol_steps.append(int(np.ceil(pp / 128.0)))
                        chol_pn.append((pp, float(pr.n)))
                        span["cholesky_chain"].append((begin, begin + ms))
                    elif name == "refit_solve":
                        cls_ms["backward_substitution"].append(ms)
            for k_, v_ in span.items():
                if v_ and min(b for b, _ in v_) >= 0:
                    windows[k_].append(max(e for _, e in v_) - min(b for b, _ in v_))
            for cx in ctxs:
                cx.enable_stage_timing(0)
    sync_all()
    env.barrier()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)
    if os.environ.get("CP_BENCH_EPOCH"):     # lets a side-car probe (tools/ubench/sidecar) find the timed region
        print("timed_region_epoch %.3f %.3f" % (epoch0, time.time()), file=sys.stderr, flush=True)
    jobs = args.steps * reps
    job_ms = elapsed / jobs * 1e3
    chunk_report = rset.chunk_report()
    # the layers of this rank are views of their contexts' result blocks (borrow_results): keep them past the runs below
    # (exchange "masks", or a rank other than 0 with "gather": a foreign layer's weights are None -- they did not travel)
    results = [(m, None if W is None else np.array(W), None if b is None else np.array(b)) for m, W, b in results]

    # ---- outside the timed region: every layer of this rank ALONE (latency, per-stage times, roofline kernel alone) ----
    per_layer = {}
    alone_g_ms, alone_g_fl, alone_g_ex = [], [], []
    alone_c_ms, alone_c_fl = [], []
    stage_by_c = {}
    for j, pr in ([] if args.profile_mode else probs.items()):
        spec = specs[rset_index[j]]
        kk = spec["k"] ** 2
        pr.ctx.enable_stage_timing(1)
        ch = [c_ for c_ in rset.chunks if j in c_["members"]][0]
        rng, mark = ch["rngs"][ch["members"].index(j)], ch["marks"][ch["members"].index(j)]
        ts = []
        for _ in range(2):
            rng_rewind(rng, mark)
            t1 = time.perf_counter()
            prune_layer(pr, spec["rank"], 1e-3, rank_tol=.1, rng=rng, mode="device")
            ts.append((time.perf_counter() - t1) * 1e3)
        st = dict(pr.ctx.last_stage_times())
        steps_cd = sum(f[2] for f in pr.fits) * spec["c"]
        per_layer[spec["name"]] = {"ms_alone": round(min(ts), 3), "kept": int(pr.refit_info.p) // kk, "fits": len(pr.fits),
                                   "cd_steps": int(steps_cd), "alpha_search_ms": round(st.get("cd_alpha_search", 0.0), 3),
                                   "cd_us_per_step": round(st.get("cd_alpha_search", 0.0) * 1e3 / max(1, steps_cd), 4),
                                   "refit_ms": round(sum(v for k_, v in st.items() if k_.startswith("refit")), 3),
                                   # which coordinate-descent kernel the width runs (include/cpmi355.h: CP_CD_FORM_*)
                                   "cd_kernel": ("one wave", "two waves", "team (one workgroup)", "multi-CU team")[
                                       pr.ctx.cd_kernel_form(spec["c"], CD_FLAGS)]}
        stage_by_c.setdefault("c%d_k%d_n%d" % (spec["c"], spec["k"], spec["n"]), st)
        if "prefactor_cholesky" in st:        # latency mode: the full Gram (P = c k k columns) factored during the search
            alone_c_ms.append(st["prefactor_cholesky"])
            alone_c_fl.append(float(spec["c"] * kk) ** 3 / 3.0)
        elif "refit_cholesky" in st:
            alone_c_ms.append(st["refit_cholesky"])
            pp_ = float(int(pr.refit_info.p))
            alone_c_fl.append(pp_ ** 3 / 3.0 + pp_ * pp_ * spec["n"])
        if "refit_gram_gemm" in st:
            alone_g_ms.append(st["refit_gram_gemm"])
            alone_g_fl.append(float(spec["N"]) * int(pr.refit_info.p) ** 2 + xty_flops(pr))
            # latency mode: the launch computed the Gram of ALL c channels during the alpha search (CP_REFIT_PRECOMPUTE)
            alone_g_ex.append(float(spec["N"]) * (spec["c"] * kk) ** 2 if ("refit_gather_normal_eq" in st or "refit_backward" in st)
                              else float(spec["N"]) * int(pr.refit_info.p) ** 2)

    # ---- N > 1, strong scaling: the bound of this mode and, in the same run, the replica throughput of the N GPUs ----
    strong_bound, replica = None, None
    if not weak:
        alone = {k_: v_["ms_alone"] for k_, v_ in per_layer.items()}
        if env.dist is not None:
            every = [None] * env.world
            env.dist.all_gather_object(every, alone)
            alone = {k_: v_ for d_ in every for k_, v_ in d_.items()}
        if alone:
            longest = max(alone, key=lambda k_: alone[k_])
            exch = float(np.mean(exch_ms)) if exch_ms else 0.0
            strong_bound = {"longest_layer_alone": longest, "longest_layer_alone_ms": round(alone[longest], 3),
                            "sum_of_layers_alone_ms": round(sum(alone.values()), 3),
                            "job_ms_lower_bound_any_gpu_count": round(alone[longest] + exch, 3),
                            "note": "one GPU already overlaps the layers of a job (job_ms at N = 1 against sum_of_layers_alone_ms); "
                                    "more GPUs cannot push ONE job below its longest layer alone + the exchange.  The >= 6x of "
                                    "north_star at 8 GPUs exists only as throughput over independent jobs: replica_throughput",
                            "row_sharding": {
                                "note": "cpmi355.shard.prune_layer_rows (the rows of one layer over two ranks: all-reduces of the "
                                        "normal equations) divides a layer's Gram and X^T Y, not its alpha search; cost test per "
                                        "layer (shard.row_shard_cost_test: GEMM at 50 TFLOP/s in a job, 150 GB/s per xGMI link); "
                                        "taken only where the saving exceeds 1.5 x the cost",
                                "layers": {s_["name"]: shard.row_shard_cost_test(s_) for s_ in specs if s_["c"] >= 256},
                                "layers_that_take_it": [s_["name"] for s_ in specs if shard.row_shard_cost_test(s_)["pays"]]}}
        if env.world > 1 and not args.profile_mode:
            # every rank prunes its OWN instance of the whole job (weak scaling, what --scaling weak times as `value`)
            if assists:
                job_set.close()
            rset.close()
            host_data.clear()
            rset = shard.ResidentLayerSet(env.local_rank, specs, lambda sp: cpjobs.synth(sp)[:3], per_stream=per_stream,
                                          flags=CD_FLAGS, borrow_results=True)
            rroots = roots = [ch["ctxs"][0] for ch in rset.chunks]     # the contexts of the sharded set are closed
            for _ in range(2):
                rset()
            for cx in rroots:
                cx.sync()
            env.barrier()
            t_r = time.perf_counter()
            rjobs = max(3, int(np.ceil(1.0 / max(job_ms * 1e-3 * min(env.world, 3), 1e-3))))
            for _ in range(rjobs):
                res_r = rset()
                shard.gather_masks(specs, res_r, env.dist)
            for cx in rroots:
                cx.sync()
            env.barrier()
            el_r = env.max_over_ranks(time.perf_counter() - t_r)
            replica = {"value": round(len(specs) * env.world * rjobs / el_r, 3), "unit": "layers/s", "job_instances": env.world,
                       "jobs_timed_per_rank": rjobs, "job_ms_per_instance": round(el_r / rjobs * 1e3, 3),
                       "note": "every GPU prunes its own instance of the whole job; the only collective is ONE uint8 all_gather of "
                               "the channel masks per job"}

    # ---- N = 1: the job with the factorisation as one launch per 128-column step (the form of rounds 4-5) and as one persistent
    # ---- launch (the default), back to back in this process: bit-identical results, what differs is the schedule
    form_ab = None
    if env.world == 1 and not args.profile_mode and not args.no_form_ab:
        lib = roots[0].lib
        lib.cp_debug_set_chol_form.argtypes = [ctypes.c_int]
        lib.cp_debug_set_chol_form.restype = ctypes.c_int
        form_ab = {}
        nj = max(6, int(np.ceil(0.5 / max(job_ms * 1e-3, 1e-3))))
        try:
            for rep in range(2):
                for name, form in (("launch_per_step", 0), ("persistent", 1)):
                    lib.cp_debug_set_chol_form(form)
                    for _ in range(3):
                        rset()
                    sync_all()
                    t_f = time.perf_counter()
                    for _ in range(nj):
                        res_f = rset()
                    sync_all()
                    form_ab.setdefault(name, []).append(round((time.perf_counter() - t_f) / nj * 1e3, 3))
                    form_ab["masks_identical"] = bool(form_ab.get("masks_identical", True) and
                                                      all(np.array_equal(a[0], b[0]) for a, b in zip(res_f, results)))
        finally:
            lib.cp_debug_set_chol_form(-1)
        form_ab = {"job_ms_launch_per_step": min(form_ab["launch_per_step"]), "job_ms_persistent": min(form_ab["persistent"]),
                   "runs": {k: v for k, v in form_ab.items() if isinstance(v, list)}, "jobs_per_run": nj,
                   "masks_identical": form_ab["masks_identical"],
                   "note": "cp_debug_set_chol_form: the same resident job, the factorisation of every refit as one launch per step / "
                           "as ONE persistent launch (the default); alternating runs, the best of two each"}

    # ---- N = 1: TWO instances of the job in flight (outside the timed region; `value` stays one job at a time).  A job alone
    # ---- leaves the chip idle under its widest layers' alpha searches (8 ms of one workgroup each) and is bound by the matrix
    # ---- pipe afterwards; a second, independent instance (another network, or another checkpoint of this one) fills the head
    pipelined = None
    if env.world == 1 and not args.profile_mode and not args.no_pipelined:
        import threading
        rset2 = shard.ResidentLayerSet(env.local_rank, [specs[i] for i in own], operands, per_stream=per_stream,
                                       flags=CD_FLAGS, borrow_results=True)
        roots2 = [ch["ctxs"][0] for ch in rset2.chunks]
        for cx in ctxs:
            cx.enable_stage_timing(0)
        rset2()
        rset()
        pj = max(4, int(np.ceil(0.6 / max(job_ms * 1e-3, 1e-3))))
        outs = [None, None]

        def loop(slot, rs):
            for _ in range(pj):
                outs[slot] = rs()

        for cx in roots + roots2:
            cx.sync()
        th = [threading.Thread(target=loop, args=(0, rset)), threading.Thread(target=loop, args=(1, rset2))]
        t_p = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        for cx in roots + roots2:
            cx.sync()
        el_p = time.perf_counter() - t_p
        same_masks = all(np.array_equal(a[0], b[0]) and np.array_equal(a[0], r[0])
                         for a, b, r in zip(outs[0], outs[1], results))
        pipelined = {"jobs_in_flight": 2, "jobs_timed": 2 * pj, "value": round(len(specs) * 2 * pj / el_p, 3), "unit": "layers/s",
                     "ms_per_job": round(el_p / (2 * pj) * 1e3, 3), "masks_identical_to_the_timed_jobs": bool(same_masks),
                     "note": "two independent instances of the whole job (own streams, contexts and host threads, operands of "
                             "their own in HBM), each running its jobs back to back; every job does all of its work.  Not `value`: "
                             "that stays ONE job at a time (job_ms = its latency)"}
        rset2.close()

    # ---- verification on rank 0 (outside the timed region) ----
    out = None
    if env.rank == 0:
        parity, werrs, recon, no_golden = True, {}, {}, []
        for spec, (idxs, newW2, newB2) in zip(specs, results):
            same, werr = golden_check(spec["name"], idxs, newW2)
            werrs[spec["name"]] = werr
            if same is not None:
                parity = parity and same and ((werr is not None and werr <= 1e-5) or (newW2 is None and args.exchange == "masks"))
            else:
                no_golden.append(spec["name"])
            if spec["layer_id"] in host_data and newW2 is not None:
                X, _, Y = host_data[spec["layer_id"]]
                Xs = X[:, idxs].reshape(spec["N"], -1).astype(np.float64)
                res = Xs @ newW2.reshape(spec["n"], -1).T + newB2 - Y
                recon[spec["name"]] = round(float(np.linalg.norm(res) / np.linalg.norm(Y)), 6)
        instances = env.world if weak else 1
        layers_per_s = len(specs) * instances * jobs / elapsed
        fl = [layer_flops(s["c"], s["n"], int(r[0].sum()) * s["k"] ** 2, N=s["N"], kk=s["k"] ** 2) for s, r in zip(specs, results)]
        by = [algorithmic_bytes(s["c"], s["n"], int(r[0].sum()) * s["k"] ** 2, N=s["N"], kk=s["k"] ** 2) for s, r in zip(specs, results)]
        alg_job, exe_job = sum(f[0] for f in fl), sum(f[1] for f in fl)
        n_sampled = (jobs + STAGE_SAMPLE - 1) // STAGE_SAMPLE       # the jobs whose stage brackets were read
        roof = roofline_object(cls_ms, g_fl, chol_fl, n_sampled, roots[0] if roots else None, PROFILE_TAG, job, windows=windows,
                               cd_steps_ns=cd_steps_ns, chol_steps=chol_steps, chol_pn=chol_pn, g_fl_bracket=g_fl_bracket, g_n=g_n)
        if roof is not None:
            roof["jobs_with_stage_brackets"] = n_sampled
        if roof is not None and alone_g_ms:
            a1 = sum(alone_g_fl) / (sum(alone_g_ms) * 1e-3) / 1e12
            roof["alone"] = {"refit_gram": {"achieved": round(a1, 3), "frac": round(a1 / F64_MFMA_PEAK_TFLOPS, 4),
                                            "avg_launch_ms": round(sum(alone_g_ms) / len(alone_g_ms), 4),
                                            "executed_tflops": round(sum(alone_g_ex) / (sum(alone_g_ms) * 1e-3) / 1e12, 3)},
                             "note": "one layer at a time = latency mode: the launch computes the Gram of ALL c channels on the "
                                     "side stream during the alpha search (executed N (c k^2)^2); achieved counts only the "
                                     "algorithmic N p^2 of the kept channels"}
            if alone_c_ms:
                a2 = sum(alone_c_fl) / (sum(alone_c_ms) * 1e-3) / 1e12
                roof["alone"]["cholesky_chain"] = {"achieved": round(a2, 3), "frac": round(a2 / F64_MFMA_PEAK_TFLOPS, 4),
                                                   "avg_ms_per_layer": round(sum(alone_c_ms) / len(alone_c_ms), 4),
                                                   "note": "p^3/3 of the matrix the layer really factored alone (single-layer calls "
                                                           "factor the FULL Gram, P = c k k columns, during the alpha search)"}
        out = {
            "metric": JOB_TEXT[job][1],
            "value": round(layers_per_s, 3), "unit": "layers/s", "n_gpus": env.world, "steps": args.steps,
            "warmup": max(1, args.warmup), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": JOB_TEXT[job][0],
                       "job_instances": instances,
                       "layers_per_job": len(specs), "jobs_per_step": reps, "jobs_timed": jobs,
                       "untimed_jobs_before": 1 + max(1, args.warmup),
                       "timed_region_s": round(elapsed, 3), "world_size": env.world, "backend": env.backend if env.dist else None,
                       "streams_per_gpu": len(rset.chunks), "layers_in_flight_per_gpu": len(own),
                       "layers_per_call": sorted({len(ch["members"]) for ch in rset.chunks}),
                       "owner_rank_of_layer": None if weak else owner,
                       "exchange_round_of_layer": exchange_rounds if (env.dist is not None and not assists) else None,
                       "row_assisted_layers": {specs[i]["name"]: {"owner": owner[i], "helper": h} for i, h in sorted(assists.items())},
                       "row_assist_timings_rank0_ms": getattr(job_set, "last_timings", None) if assists else None,
                       "parallelism": ("one job instance per GPU x%d, uint8 all_gather of the channel masks per job" % env.world)
                       if weak else ("layers of one instance sharded x%d (LPT); uint8 all_gather of the channel masks + %s%s" % (
                           env.world, {"gather": "the owners' packed (W,b) to rank 0 (point to point, exact lengths)",
                                       "allgather": "all_gather of the owners' packed (W,b) to every rank",
                                       "masks": "nothing else (weights stay with their owner)"}[args.exchange],
                           " -- in two rounds: the light layers' results travel while the heavy layers are still being pruned"
                           if exchange_rounds and len(set(exchange_rounds)) > 1 and not assists else ""))},
            "job_ms": round(job_ms, 3),
            "exchange_rank0": None if not exch_ms else dict(
                {k: (round(v, 3) if isinstance(v, float) else v) for k, v in shard.LAST_EXCHANGE_MS.items()},
                avg_total_ms=round(float(np.mean(exch_ms)), 3),
                note="host wall time of cpmi355.shard.exchange_results on rank 0 (includes waiting for the slowest rank)"),
            "mask_parity_vs_reference_golden": parity if len(no_golden) < len(specs) else None,
            "masks_identical_on_every_rank": (bool(masks_equal[0]) if (weak and env.dist is not None) else None),
            "layers_without_golden": no_golden,
            "weights_rel_frobenius_vs_reference_golden": werrs,
            "reconstruction_rel_frobenius_err": recon,
            "roofline": roof,
            "job_mfma": {"gflop_per_job_algorithmic": round(alg_job / 1e9, 1), "gflop_per_job_executed_model": round(exe_job / 1e9, 1),
                         "sustained_tflops_executed": round(exe_job / (job_ms * 1e-3) / 1e12, 2),
                         "frac_of_peak_executed": round(exe_job / (job_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS /
                                                        (1 if weak else env.world), 4),
                         "sustained_tflops_algorithmic_full_matrix_count": round(alg_job / (job_ms * 1e-3) / 1e12, 2),
                         "algorithmic_bytes_per_job": int(sum(by)),
                         "note": "executed = what the launches compute (symmetric halves of the Grams, 128-padded tiles): the "
                                 "figure to hold against the MFMA peak; algorithmic = SURVEY.md 8d's full-matrix flop count "
                                 "(2 N p^2 for a Gram whose launch executes N p^2), kept for reference only"},
            "per_layer_rank0": per_layer,
            "chunks_rank0_last_job": chunk_report,
            "stage_ms_alone_by_shape_rank0": {c: {k_: round(v, 4) for k_, v in st.items()} for c, st in stage_by_c.items()},
            "pcie_inclusive": pcie,
            "upload_and_setup_s": round(t0 - t_up0, 2),
        }
        if replica is not None:
            out["replica_throughput"] = replica
        if pipelined is not None:
            out["two_jobs_in_flight"] = pipelined
        if form_ab is not None:
            out["chol_form_ab"] = form_ab
        if not weak:
            # what sharding ONE job's layers can give: a job cannot be shorter than its longest layer alone (every layer's alpha
            # search is one serial chain, cd_team.hip), whatever the number of GPUs
            out["strong_scaling_bound"] = strong_bound
        if env.world == 1 and not args.no_cpu_baseline and not args.profile_mode:
            # bounded sample (about 20 s of CPU work): the cheapest layers of the job by the cost model
            order = sorted(specs, key=lambda s: shard.layer_cost(s["N"], s["c"], s["n"], s["k"], s["rank"]))
            small = [s for s in specs if s["c"] <= 128] + [s for s in specs if s["c"] == 256][:1] if job == "vgg16" else order[:6]
            # vgg16 (the metric's job): the port on ALL 12 layers by default (~45 s on the EPYC host), so that cpu_baseline.value is
            # layers/s of the same job and job_speedup_wall_clock is observed, not extrapolated; --cpu-sample: the five cheapest
            full = args.cpu_full or (job == "vgg16" and not args.cpu_sample)
            from .cpu_legs import cpu_baseline_object
            out["cpu_baseline"] = cpu_baseline_object(specs, specs if full else small, per_layer, job_ms, full)
    if assists and not (env.world > 1 and not args.profile_mode and not weak):     # (closed above before the replica leg)
        job_set.close()
    rset.close()
    return out

