"""benchkit.roofline -- the `roofline` object of the bench line: the two MFMA kernel classes of a job timed live with HIP events
on their launch streams, priced against the f64 MFMA peak; HBM traffic per launch from the committed rocprofv3 counter passes."""
import os

import numpy as np

from .common import F64_MFMA_PEAK_TFLOPS, N_SAMPLES, ROOT

def pmc_traffic(pattern, round_tag, per_pattern=None):
    """-> (HBM bytes per launch of the kernel whose name contains `pattern`, "<files>@<commit of the library they profiled>")
    from the committed rocprofv3 counter passes (separate --pmc runs of `bench.py --profile-mode`; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-B/lane streaming reads, WRITE_SIZE as reported).  (None, None) when absent.
    per_pattern: a kernel that is launched ONCE per layer (the refit Gram GEMM) -- the bytes are then those of all the launches
    of `pattern` that belong to one layer (the factorisation is up to four launches per layer): per launch x the ratio of the
    two kernels' dispatch counts in the same trace."""
    out, commit, disp = {}, None, {}
    pats = [pattern] if isinstance(pattern, str) else list(pattern)        # alternatives of one kernel class: bytes of the first
    pers = [per_pattern] if isinstance(per_pattern, str) else list(per_pattern or [])   # one present, dispatch counts summed
    for key, fname, scale in (("fetch", "%s_pmc_fetch_size_kb.md" % round_tag, 2.0),
                              ("write", "%s_pmc_write_size_kb.md" % round_tag, 1.0)):
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.isfile(path):
            return None, None
        best = None
        for line in open(path):
            if line.startswith("commit:"):
                commit = line.split(":", 1)[1].strip()
            for group, names in (("main", pats), ("per", pers)):
                for rank, pat in enumerate(names):
                    if pat and pat in line:
                        try:
                            cols = line.split("|")
                            disp[(key, group)] = disp.get((key, group), 0.0) + float(cols[2])
                            if group == "main" and (best is None or rank < best):
                                best = rank
                                out[key] = float(cols[3]) * 1024.0 * scale
                        except (ValueError, IndexError):
                            pass
    if len(out) != 2:
        return None, None
    total = out["fetch"] + out["write"]
    if pers:
        a, b = disp.get(("fetch", "main")), disp.get(("fetch", "per"))
        if not a or not b:
            return None, None
        total *= a / b
    return total, "profiles/%s_pmc_{fetch,write}_size_kb.md@%s" % (round_tag, commit or "unknown")


def roofline_object(cls_ms, g_fl, chol_fl, jobs, ctx0, round_tag, job, windows=None, cd_steps_ns=None, chol_steps=None, chol_pn=None,
                    g_fl_bracket=None, g_n=None):
    # g_fl: N p^2 per Gram bracket (the algorithmic Gram flops: p is derived from it); g_fl_bracket: the flops the bracket's
    # launch executed -- N p^2 + 2 N p n where the library forms X^T Y in the Gram's launch (cp_gemm_gram_xty); g_n: n per bracket
    g_exec = g_fl_bracket if g_fl_bracket else g_fl
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
    fused = bool(g_fl_bracket) and sum(g_fl_bracket) > sum(g_fl)
    gram = {"kernel": "k_gemm_tn_f64<Gram + X^T Y, refit> (G = Xs^T Xs and R = Xs^T Yc, one launch per layer)" if fused else
                      "k_gemm_tn_f64<lower, refit Gram> (G = Xs^T Xs, one launch per layer)",
            "flops_per_launch": "N p^2 (symmetric half of 2 N p^2) + 2 N p n, p = kept k k" if fused else
                                "N p^2 (symmetric half of 2 N p^2), p = kept k k",
            "achieved": round(sum(g_exec) / (sum(cls_ms["refit_gram"]) * 1e-3) / 1e12, 3),
            "avg_launch_ms": round(sum(cls_ms["refit_gram"]) / len(cls_ms["refit_gram"]), 4), "launches": len(cls_ms["refit_gram"]),
            "sum_ms_per_job": round(per_job["refit_gram"], 3), "pmc_pattern": ("k_gemm_tn_f64<3, 2,", "k_gemm_tn_f64<1, 2,")}
    chol = None
    if cls_ms["cholesky_chain"] and sum(cls_ms["cholesky_chain"]) > 0:
        chol = {"kernel": "k_chol_chain (blocked Cholesky, persistent: tile tasks off a counter, 1-4 launches per layer; p/128 steps)",
                "flops_per_launch": "per layer (all its launches): p^3 / 3 + p^2 n (the forward substitution rides along)",
                "achieved": round(sum(chol_fl) / (sum(cls_ms["cholesky_chain"]) * 1e-3) / 1e12, 3),
                "avg_launch_ms": round(sum(cls_ms["cholesky_chain"]) / len(cls_ms["cholesky_chain"]), 4),
                "launches": len(cls_ms["cholesky_chain"]), "sum_ms_per_job": round(per_job["cholesky_chain"], 3),
                "pmc_pattern": "k_chol_chain"}
        if chol_steps and sum(chol_steps) > 0:      # per 128-column step: the bracket of a layer / its p / 128 steps
            chol["avg_step_us"] = round(sum(cls_ms["cholesky_chain"]) * 1e3 / sum(chol_steps), 2)
            chol["steps_per_job"] = int(round(sum(chol_steps) / max(1, jobs)))
    top = gram if chol is None or per_job["refit_gram"] >= per_job["cholesky_chain"] else chol
    if chol is not None:
        chol["pmc_per"] = gram["pmc_pattern"]       # traffic of a factorisation = of all its launches (one Gram launch per layer)
    traffic, source = pmc_traffic(top["pmc_pattern"], round_tag, top.get("pmc_per")) if job == "vgg16" else (None, None)
    # algorithmic HBM bytes per launch, averaged over the job's launches: Gram 8 N p + 8 p^2 (the staged rows read once, the
    # Gram written once); factorisation step: G and R read once, U and Y written once, spread over the layer's p / 128 launches
    n_gram = max(1, len(g_fl))
    gram["traffic_algorithmic"] = None
    for k, flops, cls in ((gram, g_exec, "refit_gram"), (chol, chol_fl, "cholesky_chain")):
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
        t_k, src_k = pmc_traffic(k["pmc_pattern"], round_tag, k.get("pmc_per")) if job == "vgg16" else (None, None)
        if t_k is not None:
            k["traffic"] = t_k
            k["traffic_source"] = src_k
    if job == "vgg16" and g_fl:
        # p from N p^2; 8 N p + 8 p^2 per Gram launch
        ps = [np.sqrt(f / N_SAMPLES) for f in g_fl]
        ns = g_n if (fused and g_n and len(g_n) == len(ps)) else [0.0] * len(ps)      # + Yc read, R written when fused
        gram["traffic_algorithmic"] = round(float(np.mean([8.0 * N_SAMPLES * (p_ + n_) + 8.0 * p_ * (p_ + n_) for p_, n_ in zip(ps, ns)])), 1)
        if gram.get("traffic"):
            gram["traffic_ratio"] = round(gram["traffic"] / gram["traffic_algorithmic"], 2)
        if chol is not None and chol_fl:
            # per launch = per layer: G (upper half, 4 p^2 B) + R (8 p n) read once, U (4 p^2) + Y (8 p n) written once
            pn = chol_pn or [(p_, 512.0) for p_ in ps]
            chol["traffic_algorithmic"] = round(float(np.mean([8.0 * p_ * p_ + 16.0 * p_ * n_ for p_, n_ in pn])), 1)
            if chol.get("traffic"):
                chol["traffic_ratio"] = round(chol["traffic"] / chol["traffic_algorithmic"], 2)
    out = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"], "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": top["frac"], "traffic": traffic, "traffic_source": source,
           "traffic_note": "HBM bytes per launch (factorisation: per layer = all its launches): rocprofv3 --pmc FETCH_SIZE (x2, gfx950 "
                           "wide-read correction) + WRITE_SIZE passes of `bench.py --profile-mode`, committed under profiles/ (taken "
                           "at the commit named)",
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
        k.pop("pmc_per", None)
    return out

