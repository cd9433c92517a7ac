"""benchkit.line -- the ONE JSON line bench.py prints.

Everything a leg measures goes into a `detail` dict (written to bench_detail.json next to bench.py); compact_line() picks the
contract keys out of it and render() serialises them, guaranteed below MAX_LINE_BYTES: the driver's recorder reads the tail
of stdout, and a line that outgrows it is a line nobody recorded (round 5: 20.6 KB, `parsed: null`).  Nothing here measures
anything; tests/test_bench_line.py builds the line from a canned detail."""
import json

MAX_LINE_BYTES = 4096

# what every line carries (bench.py contract) -- tests/test_bench_line.py asserts them
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
CONFIG_KEYS = ("workload", "layers_per_job", "jobs_per_step", "jobs_timed", "timed_region_s", "world_size", "backend")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")

# dropped first .. last when a line would still be too long (never a contract key)
SHEDDABLE = ("patch_gather", "chol_form_ab_job_ms", "two_jobs_in_flight_layers_per_s", "owner_rank_of_layer", "replica_throughput", "r3",
             "other_workloads", "exchange", "value_conv3_block")


def _r(v, digits=4):
    """round floats for the line (None and non-floats pass through)"""
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    return v


def _short(text, limit):
    text = " ".join(str(text).split())
    return text if len(text) <= limit else text[:limit - 3] + "..."


def _kernel_entry(roof, word):
    for k in (roof or {}).get("kernels", []):
        if word in k.get("kernel", ""):
            return k
    return None


def compact_roofline(roof):
    """`roofline` of the line out of roofline_object()'s detail (None stays None)"""
    if not roof:
        return None
    out = {"bound": roof.get("bound"), "kernel": _short(roof.get("kernel", ""), 60).split(" (")[0],
           "achieved": _r(roof.get("achieved")), "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": _r(roof.get("frac")),
           "traffic": roof.get("traffic")}
    chol, gram = _kernel_entry(roof, "k_chol_"), _kernel_entry(roof, "k_gemm_tn_f64")
    top = chol if (chol is not None and out["kernel"].startswith("k_chol_")) else gram
    if top is not None:
        out["chip_level_frac"] = _r((top.get("chip_level") or {}).get("frac"))
        out["traffic_algorithmic"] = top.get("traffic_algorithmic")
        out["traffic_ratio"] = top.get("traffic_ratio")
        out["sum_ms_per_job"] = _r(top.get("sum_ms_per_job"))
        if top.get("avg_step_us") is not None:
            out["avg_step_us"] = _r(top.get("avg_step_us"))
            out["steps_per_job"] = top.get("steps_per_job")
    if gram is not None and gram is not top:
        out["gram"] = {"frac": _r(gram.get("frac")), "chip_level_frac": _r((gram.get("chip_level") or {}).get("frac")),
                       "traffic_ratio": gram.get("traffic_ratio"), "sum_ms_per_job": _r(gram.get("sum_ms_per_job"))}
    ns = (roof.get("alpha_search") or {}).get("ns_per_step_in_the_job_by_channels")
    if ns:
        out["alpha_search_ns_per_step"] = ns
    out["peak_measured"] = roof.get("peak_measured")
    out["traffic_source"] = roof.get("traffic_source")
    return out


def compact_cpu(cpu):
    if not cpu:
        return None
    out = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind")}
    out["sample"] = _short(cpu.get("sample", ""), 200)
    for k in ("cpu_model", "host_cpus", "job_seconds_cpu", "job_speedup_wall_clock", "speedup_same_layers_latency",
              "speedup_single_instance_latency", "speedup_same_convs"):
        if cpu.get(k) is not None:
            out[k] = cpu[k]
    return out


def compact_line(d):
    """the line's dict out of a leg's detail dict `d` (any workload)"""
    out = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                 "scaling", "vs_baseline", "dtype", "data")}
    cfg = d.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in CONFIG_KEYS if k in cfg or k in ("workload", "world_size")}
    out["config"]["workload"] = _short(cfg.get("workload", ""), 230)
    for k in ("passes_timed", "block_copies_in_flight", "parallelism"):
        if k in cfg and k != "parallelism":
            out["config"][k] = cfg[k]
    if cfg.get("row_assisted_layers"):
        out["config"]["row_assisted_layers"] = cfg["row_assisted_layers"]
    if d.get("job_ms") is not None:
        out["job_ms"] = d["job_ms"]
    out["roofline"] = compact_roofline(d.get("roofline"))
    jm = d.get("job_mfma") or {}
    if out["roofline"] is not None and jm.get("frac_of_peak_executed") is not None:
        out["roofline"]["job_frac_of_peak_executed"] = jm["frac_of_peak_executed"]
    out["cpu_baseline"] = compact_cpu(d.get("cpu_baseline"))
    for k in ("mask_parity_vs_reference_golden", "masks_identical_on_every_rank",
              "masks_and_alpha_chain_identical_to_the_reference_chain", "masks_identical_to_cpu_port_with_carry"):
        if d.get(k) is not None:
            out[k] = d[k]
    werr = d.get("weights_rel_frobenius_vs_reference_golden")
    if werr:
        vals = [v for v in (werr.values() if isinstance(werr, dict) else werr) if v is not None]
        if vals:
            out["weights_rel_frobenius_max"] = float("%.3g" % max(vals))
    rec = d.get("reconstruction_rel_frobenius_err")
    if rec:
        vals = [v for v in (rec.values() if isinstance(rec, dict) else rec) if v is not None]
        if vals:
            out["reconstruction_rel_frobenius_err_max"] = _r(max(vals))
    vb = d.get("value_conv3_block")
    if vb:
        out["value_conv3_block"] = {"value": vb.get("value"), "unit": vb.get("unit"), "ms_per_pass": vb.get("ms_per_pass"),
                                    "mask_parity": vb.get("mask_parity_vs_reference_golden")}
    if d.get("single_instance_layers_per_s") is not None:
        out["single_instance_layers_per_s"] = d["single_instance_layers_per_s"]
    pc = d.get("pcie_inclusive")
    if pc:
        out["pcie_inclusive_layers_per_s"] = pc.get("layers_per_s_with_h2d")
        out["pcie_inclusive_job_ms"] = pc.get("job_ms_sequential_with_h2d")
        if pc.get("x_float64"):
            out["pcie_inclusive_job_ms_x_float64"] = pc["x_float64"].get("job_ms_sequential_with_h2d")
        if pc.get("prefetched"):
            out["pcie_inclusive_job_ms_prefetched"] = pc["prefetched"].get("job_ms_sequential_with_h2d")
    sb = d.get("strong_scaling_bound")
    if sb:
        out["strong_scaling_bound_ms"] = sb.get("job_ms_lower_bound_any_gpu_count")
    tj = d.get("two_jobs_in_flight")
    if tj:
        out["two_jobs_in_flight_layers_per_s"] = tj.get("value")
    ab = d.get("chol_form_ab")
    if ab:
        out["chol_form_ab_job_ms"] = {"launch_per_step": ab.get("job_ms_launch_per_step"), "persistent": ab.get("job_ms_persistent")}
    pg = (d.get("patch_gather") or {}).get("one_launch")
    if pg:
        out["patch_gather"] = {"GBps_algorithmic": pg.get("kernel_GBps_algorithmic", pg.get("GBps_algorithmic")),
                               "frac_of_hbm_peak": pg.get("kernel_frac_of_hbm_peak", pg.get("frac_of_hbm_peak"))}
    if d.get("other_workloads"):
        out["other_workloads"] = {name: {k: v.get(k) for k in ("layers_per_s", "job_ms", "mask_parity")}
                                  for name, v in d["other_workloads"].items()}
    if d.get("r3"):
        out["r3"] = {k: d["r3"].get(k) for k in ("pass_s", "vh_s", "itq_s", "prune_s")}
    if d.get("stage_ms_per_job") and "r3" not in out:          # --workload r3's own line
        st = d["stage_ms_per_job"]
        out["r3"] = {"pass_s": _r(d.get("job_ms", 0.0) / 1e3), "vh_s": _r(st.get("spatial_decomposition (VH)", 0.0) / 1e3),
                     "itq_s": _r(st.get("channel_decomposition (ITQ)", 0.0) / 1e3),
                     "prune_s": _r(st.get("channel_pruning (dictionary)", 0.0) / 1e3)}
    # N > 1
    ex = d.get("exchange_rank0")
    if ex:
        out["exchange"] = {"bytes_sent_per_rank": ex.get("bytes_sent"), "bytes_received_per_rank": ex.get("bytes_received"),
                           "ms_rank0": ex.get("avg_total_ms"), "mode": ex.get("mode"),
                           "xgmi_model_ms": ex.get("xgmi_model_ms")}
    if d.get("replica_throughput"):
        out["replica_throughput"] = {k: d["replica_throughput"].get(k) for k in ("value", "unit", "job_ms_per_instance")}
    if cfg.get("owner_rank_of_layer") and (d.get("n_gpus") or 1) > 1:
        out["owner_rank_of_layer"] = cfg["owner_rank_of_layer"]
    out["detail"] = d.get("detail_file", "bench_detail.json")
    return out


def render(d, limit=MAX_LINE_BYTES):
    """-> the line (str, no newline), shorter than `limit` bytes: optional groups are shed, in SHEDDABLE order, should the
    compact dict still serialise too long (the contract keys never are)."""
    out = compact_line(d)
    text = json.dumps(out, separators=(", ", ": "))
    for key in SHEDDABLE:
        if len(text.encode()) < limit:
            break
        if out.pop(key, None) is not None:
            out["shed"] = out.get("shed", []) + [key]
            text = json.dumps(out, separators=(", ", ": "))
    if len(text.encode()) >= limit:
        raise ValueError("bench line is %d bytes even without its optional groups" % len(text.encode()))
    return text
