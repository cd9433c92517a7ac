"""bench.py -- conv layers pruned per second (BASELINE.json metric) on MI355X.

Workloads, all built from SURVEY.md section 8d's synthetic generator (float32 X and W2 -- exactly what the
reference's float64 arrays hold -- float64 Y), operands RESIDENT in HBM before the timed region, results (mask,
weights, bias) back on the host inside it:

--workload vgg16 (default; north_star's job, BASELINE.json configs[2] on one node)
    ONE instance of the whole-network job: the 12 conv -> conv pairs of VGG-16 with the reference's 3C-4x kept-channel
    count d_c = max(int(c / 1.15), rank) (/root/reference/lib/net.py:1309-1327, 1346-1349), N = 5000 samples per
    layer.  One job = 12 dictionary() problems (c = 64, 64, 128, 128, 256 x3, 512 x5).  With --gpus N the layers are
    sharded over the ranks (cpmi355.shard.prune_sharded: LPT assignment, ONE uint8 all_gather of the channel masks and
    ONE all_gather of the owners' packed (W, b) on RCCL) -- STRONG scaling: the same 12 layers whatever N is.  On a rank
    all its layers are in flight together (cpmi355.shard.ResidentLayerSet: a HIP stream + host thread per chunk of
    equal-width layers, the alpha searches of a chunk as the workgroups of one launch).
    A "step" is `jobs_per_step` back-to-back jobs (chosen during warm-up so that the timed region is >= 2 s);
    value = 12 * jobs / elapsed = layers/s of a single job instance, job_ms = its wall-clock.

--workload resnet50 (BASELINE.json configs[3]): the 40 selections of the released ResNet-50 2x model
    (temp/resnet-50-cp.prototxt: channel samplers in front of branch2a with c up to 2048, 3x3 and residual-aware 1x1
    consumers; cpmi355/jobs.py), N = 5000; same machinery, same JSON line.
--workload vgg16_5x (BASELINE.json configs[4]): the 10 pruned pairs of the released VGG-16 5x model
    (temp/channel_pruning.prototxt) at N = 20000 samples per layer.

--gpus N > 1 without a torch.distributed environment: bench.py launches itself under torch.distributed.run with N
ranks (one per GPU, backend "nccl" = RCCL), so `python bench.py --gpus 8` alone is a valid command.

--workload block (BASELINE.json configs[1]: the VGG-16 conv3_x block, rank = c/2)
    single_instance: the three layers of ONE block instance side by side, nothing else on the chip;
    value: replica throughput -- many independent copies of the block in flight (--inflight groups x --batch copies),
    the regime of a job with hundreds of equal layers; the line says how many really ran.

Prints ONE JSON line (rank 0).  `roofline` = the dominant MFMA kernel (the f64 Gram GEMM of the refit, symmetric half):
algorithmic flops N p^2 per launch / the HIP-event time of that launch recorded inside libcpmi355 on its launch stream
during the timed steps.  `cpu_baseline` = the CPU port of the reference path (oracle/cp_oracle.py driving scikit-learn's
own Lasso / LinearRegression: the arithmetic the reference runs) on a bounded sample of the same layers (the four with
c <= 128 and one with c = 256: about 20 s) on this box's host cores, at the BLAS thread count that is fastest on the
box (swept over 1 / 8 / 32 / all on the three smallest layers first; the CD itself is single-threaded); --cpu-full:
every layer of the job.
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")   # one hardware queue per stream (before any HIP initialisation)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))

N_SAMPLES, KSIZE = 5000, 3
# 0 = sklearn's own operation order in the CD step (the drop-in's default, lib/cfgs.py); 3 = CP_CD_RECIPROCAL | CP_CD_DELTA
# (rounding-level differences, same masks on every golden, ~4 % faster job)
CD_FLAGS = int(os.environ.get("CP_BENCH_CD_FLAGS", "0"))
F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X public FP64 matrix figure (the guide lists no f64 row); the
                              # measured v_mfma_f64_16x16x4_f64 issue rate is reported next to it
MIN_TIMED_SECONDS = 2.0

# ---- workload tables ------------------------------------------------------------------------------------------
BLOCK_LAYERS = [  # (layer_id, c, n, rank)  -- ids match tests/golden/L0[123]_*.npz
    (31, 128, 256, 64),
    (32, 256, 256, 128),
    (33, 256, 256, 128),
]
BLOCK_GOLDEN = {31: "L01_conv2_2_conv3_1", 32: "L02_conv3_1_conv3_2", 33: "L03_conv3_2_conv3_3"}

from cpmi355 import jobs as cpjobs   # noqa: E402  (workload tables + the synthetic generator; no device code)


def synth(layer_id, c, n):
    """SURVEY.md section 8d generator for the conv3_x block layers (k = 3, N = 5000, ReLU'd X)"""
    return cpjobs.synth(dict(layer_id=layer_id, N=N_SAMPLES, c=c, n=n, k=KSIZE))


def sketch_matrix(p):
    """the seeded test matrix of the sketched weight goldens (oracle/cp_oracle.py::sketch_matrix, restated)"""
    return np.random.RandomState(777).randn(int(p), 32)


def golden_check(name, idxs, newW2):
    """-> (mask identical, weight rel. Frobenius error [estimated from the sketch when the golden holds no full tensor])"""
    gpath = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.isfile(gpath):
        return None, None
    g = np.load(gpath)
    same = bool(np.array_equal(idxs, g["idxs"]))
    if not same:
        return False, None
    wm = newW2.reshape(newW2.shape[0], -1)
    if "newW2_sketch" in g.files:
        sk = wm @ sketch_matrix(wm.shape[1])
        return True, float(np.linalg.norm(sk - g["newW2_sketch"]) / np.linalg.norm(g["newW2_sketch"]))
    return True, float(np.linalg.norm(newW2 - g["newW2"]) / np.linalg.norm(g["newW2"]))


def layer_flops(c, n, pp, N=N_SAMPLES, kk=KSIZE * KSIZE):
    """(SURVEY.md 8d algorithmic flops of one dictionary() call [full-matrix counts], flops the launches execute
    [symmetric halves, 128-padded tiles])"""
    S = min(400, N // 20)
    alg = (2.0 * c * S * kk * n + 2.0 * S * n * c * c + 2.0 * S * n * c + 2.0 * N * pp * pp + 2.0 * N * pp * n
           + pp ** 3 / 3.0 + 2.0 * pp * pp * n)
    pad = lambda v, a: (v + a - 1) // a * a   # noqa: E731
    ck, P, n_pad, Np = pad(c * kk, 128), pad(pp, 128), pad(n, 128), pad(N, 16)
    tri = lambda m: m // 128 * (m // 128 + 1) // 2 * 128.0 * 128.0   # noqa: E731
    exe = (tri(ck) * 2.0 * (pad(S, 16) + n) + 2.0 * S * n * ck + tri(P) * 2.0 * Np + 2.0 * P * n_pad * Np
           + P ** 3 / 3.0 + 2.0 * P * P * n_pad)
    return alg, exe


def algorithmic_bytes(c, n, pp, N=N_SAMPLES, kk=KSIZE * KSIZE):
    """inputs once at f32 (+ the f64 Y the caller hands over) and the outputs at f64"""
    return 4.0 * (N * c * kk + N * n + n * c * kk) + 4.0 * N * n + 8.0 * (n * pp + n) + c


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([i.get("num_threads", 1) for i in threadpool_info()] + [1]))
    except Exception:
        return os.cpu_count() or 1


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_port_seconds(specs, threads=None, carry_alpha=False, masks_out=None):
    """CPU port of the reference path: seconds per layer.  specs: cpmi355.jobs spec dicts.  threads: BLAS / OpenMP
    thread limit (threadpoolctl) or None for the library default (all cores).  carry_alpha: every layer starts its search
    from the alpha the previous one ended with (cfgs.alpha, /root/reference/lib/decompose.py:491, 626-627) instead of 1e-3.
    masks_out: list that receives (idxs, alpha_out) per layer."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cp_oracle
    from threadpoolctl import threadpool_limits
    secs, alpha = [], 1e-3
    for spec in specs:
        X, W2, Y, B2 = cpjobs.synth(spec)
        X64 = X.astype(np.float64)
        np.random.seed(1234 + spec["layer_id"])
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            out = cp_oracle.dictionary_oracle(X64, W2, Y, spec["rank"], B2, alpha_in=alpha if carry_alpha else 1e-3,
                                              lasso="sklearn", ls="sklearn")
            secs.append(time.perf_counter() - t0)
        if carry_alpha:
            alpha = out[3]
        if masks_out is not None:
            masks_out.append((out[0], out[3]))
    return secs


def cpu_best_threads(specs):
    """-> (thread count that minimises the port's time on `specs`, {threads: seconds})"""
    ncpu = os.cpu_count() or 1
    sweep = {}
    for t in sorted({1, min(8, ncpu), min(32, ncpu), ncpu}):
        sweep[t] = round(sum(cpu_port_seconds(specs, threads=t)), 3)
    return min(sweep, key=lambda t: sweep[t]), sweep


def cpu_baseline_object(specs, sample, per_layer, job_ms, full, carry_alpha=False, masks_out=None):
    """cpu_baseline of the JSON line: the port on `sample` at the best BLAS thread count of this box."""
    best, sweep = cpu_best_threads(sample[:3])
    secs = cpu_port_seconds(sample, threads=best, carry_alpha=carry_alpha, masks_out=masks_out)
    gpu_ms_same = sum(per_layer[s["name"]]["ms_alone"] for s in sample if s["name"] in per_layer)
    out = {"value": round(len(sample) / sum(secs), 4), "unit": "layers/s", "cores": int(best), "kind": "port",
           "sample": "%s of the job's %d layers (%s), one pass, sklearn Lasso (single-threaded CD) + LinearRegression/gelsd "
                     "with %d BLAS threads (the fastest of the sweep): %.1f s total, per layer %s s" % (
                         "all" if full else "%d" % len(sample), len(specs), ", ".join(s["name"][:3] for s in sample), best,
                         sum(secs), [round(x, 2) for x in secs]),
           "blas_thread_sweep_s": {"layers": [s["name"][:3] for s in sample[:3]], "seconds_by_threads": sweep},
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    if gpu_ms_same > 0:
        out["gpu_ms_same_layers_one_at_a_time"] = round(gpu_ms_same, 2)
        out["speedup_same_layers_latency"] = round(sum(secs) * 1e3 / gpu_ms_same, 1)
    if full:
        out["job_seconds_cpu"] = round(sum(secs), 2)
        out["job_speedup_wall_clock"] = round(sum(secs) * 1e3 / job_ms, 1)
    return out


# ==================================================================================================================
# distributed plumbing
# ==================================================================================================================
class Env:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = os.environ.get("CP_BENCH_DIST_BACKEND", "nccl")   # "gloo": several ranks on ONE GPU (flow test)
        self.dist = None
        self.torch = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                self.local_rank = self.local_rank % max(1, torch.cuda.device_count())
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group(self.backend)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_int(self, value):
        if self.dist is None:
            return int(value)
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.broadcast(t, src=0)
        return int(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def git_head():
    try:
        import subprocess
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def pmc_traffic(pattern, round_tag):
    """-> (HBM bytes per launch of the kernel whose name contains `pattern`, "<files>@<commit of the library they profiled>")
    from the committed rocprofv3 counter passes (separate --pmc runs of `bench.py --profile-mode`; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-B/lane streaming reads, WRITE_SIZE as reported).  (None, None) when absent."""
    out, commit = {}, None
    for key, fname, scale in (("fetch", "%s_pmc_fetch_size_kb.md" % round_tag, 2.0),
                              ("write", "%s_pmc_write_size_kb.md" % round_tag, 1.0)):
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.isfile(path):
            return None, None
        for line in open(path):
            if line.startswith("commit:"):
                commit = line.split(":", 1)[1].strip()
            if pattern in line:
                try:
                    out[key] = float(line.split("|")[3]) * 1024.0 * scale
                except (ValueError, IndexError):
                    pass
    if len(out) != 2:
        return None, None
    return out["fetch"] + out["write"], "profiles/%s_pmc_{fetch,write}_size_kb.md@%s" % (round_tag, commit or "unknown")


def roofline_object(cls_ms, g_fl, chol_fl, jobs, ctx0, round_tag, job, windows=None, cd_steps_ns=None):
    """`roofline` of the JSON line.  The kernel classes of the job's MFMA work, each timed live with HIP events on its launch
    stream during the timed jobs (cp_enable_stage_timing mode 2): the refit Gram GEMM (one launch per layer) and the
    factorisation chain (the Cholesky step launches of a layer, with the forward substitution riding along).  The one
    with the larger sum over the job is THE roofline kernel; both are listed under `kernels`, next to the two
    latency-bound chains (alpha search, backward substitution) whose sums say where the rest of the time goes."""
    if not cls_ms["refit_gram"] or sum(cls_ms["refit_gram"]) <= 0:
        return None
    # the best of three: one reading in a while comes out at half the rate (145 cycles per instruction at the full clock: the
    # launch shared the chip with the tail of something else), and the ceiling is what the pipe CAN issue
    probe_tf, ghz, cyc = max((ctx0.probe_mfma_f64_clock() for _ in range(3)), key=lambda t: t[0])
    per_job = {k: sum(v) / max(1, jobs) for k, v in cls_ms.items()}
    gram = {"kernel": "k_gemm_tn_f64<lower, refit Gram> (G = Xs^T Xs, one launch per layer)",
            "flops_per_launch": "N p^2 (symmetric half of 2 N p^2), p = kept k k",
            "achieved": round(sum(g_fl) / (sum(cls_ms["refit_gram"]) * 1e-3) / 1e12, 3),
            "avg_launch_ms": round(sum(cls_ms["refit_gram"]) / len(cls_ms["refit_gram"]), 4), "launches": len(cls_ms["refit_gram"]),
            "sum_ms_per_job": round(per_job["refit_gram"], 3), "pmc_pattern": "k_gemm_tn_f64<1, 2,"}
    chol = None
    if cls_ms["cholesky_chain"] and sum(cls_ms["cholesky_chain"]) > 0:
        chol = {"kernel": "k_chol_step (blocked Cholesky, one launch per 128-column step; per layer: p/128 launches)",
                "flops_per_launch": "per layer: p^3 / 3 + p^2 n (the forward substitution rides in the same launches)",
                "achieved": round(sum(chol_fl) / (sum(cls_ms["cholesky_chain"]) * 1e-3) / 1e12, 3),
                "avg_launch_ms": round(sum(cls_ms["cholesky_chain"]) / len(cls_ms["cholesky_chain"]), 4),
                "launches": len(cls_ms["cholesky_chain"]), "sum_ms_per_job": round(per_job["cholesky_chain"], 3),
                "pmc_pattern": "k_chol_step", "avg_launch_note": "one bracket = all step launches of a layer"}
    top = gram if chol is None or per_job["refit_gram"] >= per_job["cholesky_chain"] else chol
    traffic, source = pmc_traffic(top["pmc_pattern"], round_tag) if job == "vgg16" else (None, None)
    # algorithmic HBM bytes per launch, averaged over the job's launches: Gram 8 N p + 8 p^2 (the staged rows read once, the
    # Gram written once); factorisation step: G and R read once, U and Y written once, spread over the layer's p / 128 launches
    n_gram = max(1, len(g_fl))
    gram["traffic_algorithmic"] = None
    for k, flops, cls in ((gram, g_fl, "refit_gram"), (chol, chol_fl, "cholesky_chain")):
        if k is None:
            continue
        k["frac"] = round(k["achieved"] / F64_MFMA_PEAK_TFLOPS, 4)
        k["frac_of_measured_peak"] = round(k["achieved"] / probe_tf, 4)
        # what the chip does, not what one stream sees: the flops of all the concurrent brackets of this class in a job
        # divided by the wall window they span (cp_last_stage_spans: one clock for all the layers' streams)
        w = (windows or {}).get(cls) or []
        if w and sum(w) > 0:
            tf = sum(flops) / (sum(w) * 1e-3) / 1e12
            k["chip_level"] = {"achieved": round(tf, 3), "frac": round(tf / F64_MFMA_PEAK_TFLOPS, 4),
                               "window_ms_per_job": round(sum(w) / len(w), 3),
                               "note": "flops of all the layers' brackets of this class in a job / the wall window from the first "
                                       "begin to the last end (the brackets of different layers overlap)"}
        t_k, src_k = pmc_traffic(k["pmc_pattern"], round_tag) if job == "vgg16" else (None, None)
        if t_k is not None:
            k["traffic"] = t_k
            k["traffic_source"] = src_k
    if job == "vgg16" and g_fl:
        # p from N p^2; 8 N p + 8 p^2 per Gram launch
        ps = [np.sqrt(f / N_SAMPLES) for f in g_fl]
        gram["traffic_algorithmic"] = round(float(np.mean([8.0 * N_SAMPLES * p_ + 8.0 * p_ * p_ for p_ in ps])), 1)
        if gram.get("traffic"):
            gram["traffic_ratio"] = round(gram["traffic"] / gram["traffic_algorithmic"], 2)
        if chol is not None and chol_fl:
            # per layer: G (upper half, 4 p^2 B) + R (8 p n) read, U (4 p^2) + Y (8 p n) written; per launch: / (p / 128)
            per_launch = [(8.0 * p_ * p_ + 16.0 * p_ * 512.0) / max(1.0, np.ceil(p_ / 128.0)) for p_ in ps]
            chol["traffic_algorithmic"] = round(float(np.sum([(8.0 * p_ * p_ + 16.0 * p_ * 512.0) for p_ in ps]) /
                                                      max(1.0, np.sum([np.ceil(p_ / 128.0) for p_ in ps]))), 1)
            if chol.get("traffic"):
                chol["traffic_ratio"] = round(chol["traffic"] / chol["traffic_algorithmic"], 2)
            del per_launch
    out = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"], "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": top["frac"], "traffic": traffic, "traffic_source": source,
           "traffic_note": "HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 wide-read correction) + WRITE_SIZE "
                           "passes of `bench.py --profile-mode`, committed under profiles/ (taken at the commit named)",
           "dominant_by": "sum of launch time per job among the MFMA kernels, HIP events on the launch streams during the timed jobs",
           "avg_launch_ms": top["avg_launch_ms"], "launches": top["launches"], "flops_per_launch": top["flops_per_launch"],
           "peak_nominal": F64_MFMA_PEAK_TFLOPS,
           "peak_measured": round(probe_tf, 2), "frac_of_measured_peak": top["frac_of_measured_peak"],
           "effective_ghz": round(ghz, 3), "cycles_per_mfma_measured": round(cyc, 1),
           "peak_note": "peak = 78.6 TFLOP/s, AMD's FP64 matrix figure (64 cycles per v_mfma_f64_16x16x4_f64 and SIMD at 2.4 GHz); "
                        "peak_measured = back-to-back MFMAs with VGPR accumulators (the form every kernel of the library "
                        "uses), 2 waves per SIMD x 8 accumulators, stamped with s_memtime / s_memrealtime in this run: one "
                        "instruction per 64-69 cycles at the full clock (effective_ghz).  With AccVGPR accumulators the same "
                        "instruction issues once per ~107 cycles (46.7 TFLOP/s): the figure quoted as the ceiling until the "
                        "middle of round 4 (profiles/r04_gemm_probe.md, r04_mfma_clock.md)",
           "kernels": [k for k in (gram, chol) if k is not None],
           "latency_bound_chains_ms_per_job": {"alpha_search (one workgroup-team per layer)": round(per_job["alpha_search"], 3),
                                               "backward_substitution (banded)": round(per_job["backward_substitution"], 3)},
           "alpha_search": {"ns_per_step_in_the_job_by_channels": {str(c_): round(float(np.mean(v_)), 1)
                                                                   for c_, v_ in sorted((cd_steps_ns or {}).items())},
                            "cycles_per_step_in_the_job_by_channels": {str(c_): round(float(np.mean(v_)) * ghz, 1)
                                                                       for c_, v_ in sorted((cd_steps_ns or {}).items())},
                            "note": "bracket of the whole search of a layer / (sum of n_iter over its fits x channels): one "
                                    "coordinate step of scikit-learn's Gram-form recurrence; cycles at effective_ghz"},
           "note": "brackets are stream time of a layer while the other layers of the job share the CUs; sums over the "
                   "layers of a job exceed job_ms because the layers overlap"}
    for k in out["kernels"]:
        k.pop("pmc_pattern", None)
    return out


# ==================================================================================================================
# workload: vgg16 (the north_star job)
# ==================================================================================================================
JOB_TEXT = {
    "vgg16": ("vgg16: ONE instance of the whole-network job = the 12 conv->conv pairs of VGG-16, kept channels "
              "d_c = int(c/1.15) (the reference's 3C-4x table), N=5000 samples/layer, k=3; 1 job = 12 dictionary() calls; "
              "1 step = jobs_per_step back-to-back jobs", "conv layers pruned/sec (VGG-16 4x, 5k samples)"),
    "resnet50": ("resnet50: ONE instance of the ResNet-50 2x job = the 40 selections of the released model "
                 "(temp/resnet-50-cp.prototxt): 16 channel samplers in front of branch2a (c = 64..2048, 1x1), 8 branch2a->"
                 "branch2b (3x3), 16 branch2b->branch2c (1x1, residual-aware target, no ReLU), N=5000 samples/layer; "
                 "1 step = jobs_per_step back-to-back jobs", "conv layers pruned/sec (ResNet-50 2x, 5k samples)"),
    "vgg16_5x": ("vgg16_5x: ONE instance of the VGG-16 5x job = the 10 pruned conv->conv pairs of the released model "
                 "(temp/channel_pruning.prototxt kept counts 24,22,41,51,108,89,111,184,276,228), N=20000 samples/layer, "
                 "k=3; 1 step = jobs_per_step back-to-back jobs", "conv layers pruned/sec (VGG-16 5x, 20k samples)"),
}


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
                shard.LAST_EXCHANGE_MS.update(total=(time.perf_counter() - t_x) * 1e3, bytes_sent=sum(s["c"] for s in specs))
            return res
        # N > 1: the results of the light layers are exchanged while the heavy ones are still being pruned (shard.plan_rounds)
        return shard.prune_sharded(specs, compute_many=job_set, dist=env.dist, owner=owner,
                                   staging="device" if env.dist is not None else None,
                                   rounds=exchange_rounds if (env.dist is not None and not assists) else None)

    def sync_all():
        for cx in roots:
            cx.sync()

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
    cls_ms = {"alpha_search": [], "refit_gram": [], "cholesky_chain": [], "backward_substitution": []}   # per launch / bracket, in the job
    chol_fl = []
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
                        g_fl.append(float(pr.N) * int(pr.refit_info.p) ** 2)
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
    results = [(m, np.array(W), np.array(b)) for m, W, b in results]

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
            alone_g_fl.append(float(spec["N"]) * int(pr.refit_info.p) ** 2)
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
                parity = parity and same and werr is not None and werr <= 1e-5
            else:
                no_golden.append(spec["name"])
            if spec["layer_id"] in host_data:
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
                               cd_steps_ns=cd_steps_ns)
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
                       if weak else ("layers of one instance sharded x%d (LPT), masks all_gather + all_gather of the "
                                     "packed (W,b) -- in two rounds: the light layers' results travel while the heavy "
                                     "layers are still being pruned" % env.world)},
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
            out["cpu_baseline"] = cpu_baseline_object(specs, specs if full else small, per_layer, job_ms, full)
    if assists and not (env.world > 1 and not args.profile_mode and not weak):     # (closed above before the replica leg)
        job_set.close()
    rset.close()
    return out


# measured ms of one layer alone by channel count (profiles/r02_*): the LPT costs of the vgg16 job
VGG16_COST_MS = {64: 1.5, 128: 3.0, 256: 6.8, 512: 15.5}
PROFILE_TAG = "r05"      # profiles/<tag>_pmc_{fetch,write}_size_kb.md feed roofline.traffic


# ==================================================================================================================
# workload: block (configs[1], conv3_x) -- single instance + replica throughput
# ==================================================================================================================
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


def bench_patch_gather(device, C=256, H=56, W=56, B=10, P=10, nb=50, k=3, pad=1, reps=20):
    """a1 (Net.extract_XY, lib/net.py:534-684): the sampled-point im2col of SURVEY.md 8d's gather workload -- B = 10 images,
    C = 256 channels of 56 x 56, 10 sampled points per batch, 50 batches => N = 5000 rows of C*k*k floats (ReLU fused).
    The feature maps of all batches are resident in HBM ([nb, B, C, H, W] float32 = 1.6 GB); timed: (a) ONE launch over
    all batches (cp_patch_gather_batches), (b) one cp_patch_gather call per batch as the facade issues them while the
    provider's forward passes run.  Algorithmic bytes = the rows written + the same bytes read (8 N C k^2)."""
    import cpmi355
    ctx = cpmi355.Context(device)
    try:
        rs = np.random.RandomState(7)
        one = rs.randn(B, C, H, W).astype(np.float32)
        fm = ctx.empty(nb * one.nbytes)
        for b in range(nb):           # the same batch image nb times: contents are irrelevant to a gather's speed
            ctx._check(ctx.lib.cp_memcpy_h2d(ctx.h, fm.ptr + b * one.nbytes, one.ctypes.data, one.nbytes), "cp_memcpy_h2d")
        xs = rs.randint(0, H, nb * P).astype(np.int32)
        ys = rs.randint(0, W, nb * P).astype(np.int32)
        N = nb * P * B
        out = ctx.empty(N * C * k * k * 4)
        alg = 8.0 * N * C * k * k
        res = {}
        for name in ("one_launch", "per_batch_calls"):
            def run():
                if name == "one_launch":
                    ctx.patch_gather_batches(fm, nb, B, C, H, W, xs, ys, P, k, pad, 1, True, out)
                else:
                    for b in range(nb):
                        ctx.patch_gather(fm.ptr + b * one.nbytes, B, C, H, W, xs[b * P:(b + 1) * P],
                                         ys[b * P:(b + 1) * P], k, pad, 1, True, out, b * P * B)
            run()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            ctx.sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            res[name] = {"ms": round(ms, 4), "GBps_algorithmic": round(alg / ms / 1e6, 1),
                         "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 8.0e12, 4)}
            if name == "one_launch":     # the kernel alone, HIP events on the launch stream (the call also uploads the points)
                ctx.enable_stage_timing(1)
                kms = []
                for _ in range(5):
                    run()
                    ctx.sync()
                    kms.append(dict(ctx.last_stage_times()).get("gather_kernel", 0.0))
                ctx.enable_stage_timing(0)
                kms = float(np.median(kms))
                if kms > 0:
                    res[name].update(kernel_ms=round(kms, 4), kernel_GBps_algorithmic=round(alg / kms / 1e6, 1),
                                     kernel_frac_of_hbm_peak=round(alg / (kms * 1e-3) / 8.0e12, 4))
        res["workload"] = "B=%d C=%d %dx%d k=%d pad=%d, %d points x %d batches: N=%d rows, %.1f MB written" % (
            B, C, H, W, k, pad, P, nb, N, alg / 2e6)
        res["note"] = ("HBM-bound gather: every sampled k-wide run of a channel row costs a whole 64-byte fabric request "
                       "(3 x 64 B fetched per 36 B used at k = 3), so the algorithmic rate is bounded near "
                       "8 TB/s x 72 / (192 + 36) = 2.5 TB/s; see DESIGN.md")
        return res
    finally:
        ctx.close()


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
        cpu_masks = []
        out["cpu_baseline"] = cpu_baseline_object(specs, specs, {}, job_ms, True, carry_alpha=True, masks_out=cpu_masks)
        out["masks_identical_to_cpu_port_with_carry"] = bool(all(np.array_equal(g[0], c_[0]) and g[1] == c_[1]
                                                                  for g, c_ in zip(res, cpu_masks)))
    for pr in probs:
        pr.free()
    ctx.close()
    return out


# ==================================================================================================================
# workload: r3 -- the three steps Net.R3 runs per conv of VGG-16
# ==================================================================================================================
VGG16_CONVS = [("conv1_1", 3, 64), ("conv1_2", 64, 64), ("conv2_1", 64, 128), ("conv2_2", 128, 128), ("conv3_1", 128, 256),
               ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512),
               ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512)]
# /root/reference/lib/net.py:1309-1327: the 3C-4x ranks (conv5_x as listed, the others x 4 / dic.keep with dic.keep = 3)
R3_RANK = {"conv1_2": 17, "conv2_1": 37, "conv2_2": 47, "conv3_1": 83, "conv3_2": 89, "conv3_3": 106, "conv4_1": 175,
           "conv4_2": 192, "conv4_3": 227, "conv5_1": 398, "conv5_2": 390, "conv5_3": 379}
R3_PRUNED = ("conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv4_1", "conv4_2")   # alldic + pooldic (net.py:1307-1308)


def r3_plan():
    """Per conv of the reference's R3 loop (net.py:1339-1459, conv1_2 .. conv5_3): the shapes VH_decompose, ITQ_decompose and
    dictionary_kernel see -- input channels already reduced when the conv was the consumer of an earlier pruning."""
    plan, kept_in = [], {}
    for i, (name, c, n) in enumerate(VGG16_CONVS[1:], start=1):
        rank = R3_RANK[name] if name.startswith("conv5") else int(R3_RANK[name] * 4.0 / 3.0)
        d_c = max(int(n / 1.15), rank)
        step = dict(name=name, c=kept_in.get(name, c), n=n, rank=rank, d_c=d_c, prune=None)
        if name in R3_PRUNED and i + 1 < len(VGG16_CONVS):
            nxt, _, n_next = VGG16_CONVS[i + 1]
            step["prune"] = dict(consumer=nxt, n_next=n_next)
            kept_in[nxt] = d_c
        plan.append(step)
    return plan


def r3_operands(step, seed, N=N_SAMPLES):
    rs = np.random.RandomState(seed)
    c, n = step["c"], step["n"]
    X = np.maximum(rs.randn(N, c, 3, 3), 0).astype(np.float32)
    W = (rs.randn(n, c, 3, 3) * 0.05).astype(np.float32)
    Y = X.reshape(N, -1).astype(np.float64) @ W.reshape(n, -1).T.astype(np.float64) + 0.01 * rs.randn(N, n)
    feat = Y + 0.02 * rs.randn(N, n)          # the response of the spatially decomposed conv at the same points
    out = dict(X=X, W=W, Y=Y, feat=feat)
    if step["prune"]:
        n2 = step["prune"]["n_next"]
        Xo = np.maximum(rs.randn(N, n, 3, 3), 0).astype(np.float32)
        W2 = (rs.randn(n2, n, 3, 3) * 0.05).astype(np.float32)
        out.update(Xo=Xo, W2=W2, Y2=Xo.reshape(N, -1).astype(np.float64) @ W2.reshape(n2, -1).T.astype(np.float64) + 0.01 * rs.randn(N, n2))
    return out


def bench_r3(args, env):
    """`--workload r3`: what Net.R3 (/root/reference/lib/net.py:1292-1471) runs per conv of VGG-16 -- spatial decomposition
    (VH_decompose with the ReLU-aware refit of H: 50 alternations), channel decomposition (ITQ_decompose: 50 alternations, each
    a rank-truncated SVD) and, for the 7 convs of alldic / pooldic, channel pruning against the next conv (dictionary) -- at
    the reference's 3C-4x ranks, N = 5000 sampled points per conv, through the drop-in functions of lib/decompose.py from
    host arrays.  Synthetic per-conv operands; the forward passes that re-extract features between the steps (Caffe in the
    reference, a torch provider in lib/provider.py) are not part of the timed work."""
    import cpmi355
    import lib.cfgs as cfgs
    import lib.decompose as D
    ctx = cpmi355.default_context(env.local_rank)
    plan = r3_plan()
    data = [r3_operands(st, 4000 + i) for i, st in enumerate(plan)]

    def one_pass(record=None):
        cfgs.alpha = 1e-3
        for i, (st, d) in enumerate(zip(plan, data)):
            np.random.seed(2000 + i)
            t0 = time.perf_counter()
            V, H, VHr, b = D.VH_decompose(d["W"], rank=st["rank"], DEBUG=True, X=d["X"], Y=d["Y"])
            ctx.sync()
            t1 = time.perf_counter()
            D.ITQ_decompose(d["feat"], d["Y"], H, st["rank"], bias=b, DEBUG=0, Wr=VHr)
            ctx.sync()
            t2 = time.perf_counter()
            if st["prune"]:
                D.dictionary(d["Xo"].astype(np.float64, copy=False), d["W2"], d["Y2"], rank=st["d_c"])
                ctx.sync()
            t3 = time.perf_counter()
            if record is not None:
                r = record.setdefault(st["name"], dict(vh=[], itq=[], prune=[]))
                r["vh"].append((t1 - t0) * 1e3)
                r["itq"].append((t2 - t1) * 1e3)
                r["prune"].append((t3 - t2) * 1e3)

    for _ in range(max(1, min(args.warmup, 2))):
        one_pass()
    rec = {}
    steps = max(1, min(args.steps, 3))
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(rec)
    elapsed = time.perf_counter() - t0
    job_ms = elapsed / steps * 1e3
    per = {k: {"c": st["c"], "n": st["n"], "rank": st["rank"], "d_c": st["d_c"] if st["prune"] else None,
               "vh_ms": round(min(rec[k]["vh"]), 2), "itq_ms": round(min(rec[k]["itq"]), 2),
               "prune_ms": round(min(rec[k]["prune"]), 2) if st["prune"] else None}
           for k, st in ((st["name"], st) for st in plan)}
    out = {"metric": "conv layers decomposed + pruned/sec (VGG-16 3C 4x steps of Net.R3, 5k samples)",
           "value": round(len(plan) * steps / elapsed, 3), "unit": "layers/s", "n_gpus": 1, "steps": steps,
           "warmup": max(1, min(args.warmup, 2)), "ms_per_step": round(job_ms, 2), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "r3: the 12 convs conv1_2 .. conv5_3 of VGG-16, per conv VH_decompose (rank-truncated SVD + 50 ReLU-aware "
                                  "refits) -> ITQ_decompose (50 alternations) -> dictionary() for the 7 convs the reference prunes (alldic + "
                                  "pooldic), 3C-4x ranks of /root/reference/lib/net.py:1309-1327, N = 5000; one after another as R3 does; "
                                  "1 step = 1 pass over the 12 convs",
                      "layers_per_job": len(plan)},
           "job_ms": round(job_ms, 2),
           "stage_ms_per_job": {"spatial_decomposition (VH)": round(sum(v["vh_ms"] for v in per.values()), 2),
                                "channel_decomposition (ITQ)": round(sum(v["itq_ms"] for v in per.values()), 2),
                                "channel_pruning (dictionary)": round(sum(v["prune_ms"] or 0.0 for v in per.values()), 2)},
           "per_conv": per, "roofline": None,
           "note": "host-inclusive: every call starts from NumPy arrays, as Net.R3 hands them over; latency-bound by the Jacobi "
                   "sweeps of the SVDs (svd_jacobi.hip) -- no roofline kernel is named for this workload"}
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cp_oracle
        from threadpoolctl import threadpool_limits
        sample = [0, 1]                               # conv1_2, conv2_1: about 20-40 s of CPU work at 8 BLAS threads
        secs = {}
        with threadpool_limits(limits=8):
            for i in sample:
                st, d = plan[i], data[i]
                np.random.seed(2000 + i)
                t0 = time.perf_counter()
                V, H, VHr, b = cp_oracle.vh_decompose_oracle(d["W"].astype(np.float64), rank=st["rank"], X=d["X"].astype(np.float64), Y=d["Y"])
                t1 = time.perf_counter()
                cp_oracle.itq_decompose_oracle(d["feat"], d["Y"], H, st["rank"], bias=b, Wr=VHr)
                t2 = time.perf_counter()
                if st["prune"]:
                    cp_oracle.dictionary_oracle(d["Xo"].astype(np.float64), d["W2"], d["Y2"], st["d_c"], alpha_in=1e-3,
                                                lasso="sklearn", ls="sklearn")
                t3 = time.perf_counter()
                secs[st["name"]] = dict(vh_s=round(t1 - t0, 2), itq_s=round(t2 - t1, 2), prune_s=round(t3 - t2, 2))
        cpu_s = sum(sum(v.values()) for v in secs.values())
        gpu_ms = sum(per[plan[i]["name"]]["vh_ms"] + per[plan[i]["name"]]["itq_ms"] + (per[plan[i]["name"]]["prune_ms"] or 0.0) for i in sample)
        out["cpu_baseline"] = {"value": round(len(sample) / cpu_s, 4), "unit": "layers/s", "cores": 8, "kind": "port",
                               "sample": "the first %d convs (%s): scipy gesvd / sklearn restatement of the three steps (oracle/cp_oracle.py), "
                                         "8 BLAS threads: %.1f s" % (len(sample), ", ".join(plan[i]["name"] for i in sample), cpu_s),
                               "per_conv_s": secs, "gpu_ms_same_convs": round(gpu_ms, 2),
                               "speedup_same_convs": round(cpu_s * 1e3 / gpu_ms, 1), "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    return out


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` with no torch.distributed environment: run this same command under
    torch.distributed.run, one rank per GPU on this node (rendezvous on 127.0.0.1)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("vgg16", "resnet50", "vgg16_5x", "block", "r3"),
                    default=os.environ.get("CP_BENCH_WORKLOAD", "vgg16"))
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("CP_BENCH_SCALING", "strong"),
                    help="N > 1: strong (default) = ONE job instance, its layers sharded over the GPUs (BASELINE configs[2]); the line "
                         "also carries replica_throughput (one instance per GPU) and strong_scaling_bound; weak = replicas as `value`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU port on every layer of the job (the default for vgg16: ~45 s)")
    ap.add_argument("--cpu-sample", action="store_true", help="vgg16: CPU port on the five cheapest layers only (~5 s)")
    ap.add_argument("--sequential-alpha", action="store_true",
                    help="vgg16 on one GPU, the reference's own order: layer after layer, every search starting from the alpha "
                         "the previous layer ended with (cfgs.alpha carry, /root/reference/lib/decompose.py:491, 626-627)")
    ap.add_argument("--no-block", action="store_true", help="vgg16: skip the conv3_x single-instance figures")
    ap.add_argument("--precompute-heaviest", type=int, default=None,
                    help="layers whose full normal equations are computed under their alpha search (default: the library's 2)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the two-jobs-in-flight leg (N = 1)")
    ap.add_argument("--no-row-assist", action="store_true",
                    help="N > 1, strong: never split a layer's refit rows over its owner and a helper rank (shard.plan_assists)")
    ap.add_argument("--no-exchange-rounds", action="store_true",
                    help="N > 1, strong: ONE exchange after all layers instead of the light layers' results travelling early")
    ap.add_argument("--no-gather", action="store_true", help="skip the sampled-point im2col (extract_XY) measurement")
    ap.add_argument("--no-pcie-f64", action="store_true", help="skip the float64-X variant of the PCIe-inclusive pass")
    ap.add_argument("--profile-mode", action="store_true",
                    help="only whole jobs (1 + warmup + steps x jobs_per_step of them), nothing else on the GPU: for rocprofv3")
    ap.add_argument("--per-stream", type=int, default=int(os.environ.get("CP_BENCH_PER_STREAM", "0")),
                    help="equal-width layers per stream / cp_prune_layers call (0: 1, resnet50: 2)")
    ap.add_argument("--jobs-per-step", type=int, default=0, help="fixed jobs per step (0 = fill >= 2 s)")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("CP_BENCH_INFLIGHT", "6")),
                    help="block: worker groups (3 HIP streams each) running passes over the block concurrently")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("CP_BENCH_BATCH", "8")),
                    help="block: block copies a worker group prunes per cp_prune_layers call")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    env = Env()
    if env.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the torch.distributed environment has WORLD_SIZE=%d" % (args.gpus, env.world))
    if args.sequential_alpha:
        if env.world != 1:
            raise SystemExit("bench.py: --sequential-alpha is a one-GPU mode (every layer needs the previous layer's alpha)")
        out = bench_sequential_alpha(args, env)
    elif args.workload == "r3":
        if env.world != 1:
            raise SystemExit("bench.py: --workload r3 is a one-GPU workload (R3 is a sequential loop over the convs)")
        out = bench_r3(args, env)
    elif args.workload != "block":
        out = bench_job(args, env, args.workload)
        if out is not None and not args.profile_mode and env.world == 1 and not args.sequential_alpha:
            if args.workload == "vgg16" and not args.no_block:
                single, group = block_single_instance(env.local_rank)
                close_workers(group)
                out["conv3_block_single_instance"] = single
                # north_star's headline shape (BASELINE.json configs[1]: the conv3_x block, rank = c / 2, 5000 samples, one
                # MI355X) as a first-class number next to `value` (configs[2] on one GPU): layers/s of ONE instance of the
                # block, its three layers side by side, masks checked against the reference goldens L01..L03
                out["value_conv3_block"] = {"value": single["layers_per_s"], "unit": "layers/s", "ms_per_pass": single["ms_per_pass"],
                                            "mask_parity_vs_reference_golden": single["mask_parity_vs_reference_golden"],
                                            "workload": "conv3_x block (conv2_2->conv3_1 c=128, conv3_1->conv3_2 and "
                                                        "conv3_2->conv3_3 c=256; n=256, k=3, rank=c/2), N=5000, one instance"}
            if not args.no_gather:
                out["patch_gather"] = bench_patch_gather(env.local_rank)
    else:
        out = bench_block(args, env)
    if env.rank == 0 and out is not None:
        print(json.dumps(out), flush=True)
    env.close()
    sys.stdout.flush()
    sys.stderr.flush()


if __name__ == "__main__":
    main()
