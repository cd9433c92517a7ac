"""The ONE line bench.py prints (benchkit/line.py): built from canned detail dicts -- the full lines earlier rounds printed,
kept under profiles/ -- it must stay below 4 KB, parse, and carry the contract keys.  Round 5's line was 20.6 KB and the
driver recorded `parsed: null`."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchkit import line   # noqa: E402

CANNED = ["r05_bench_vgg16.json", "r05_bench_resnet50.json", "r05_bench_vgg16_5x.json", "r05_bench_r3.json",
          "r05_bench_vgg16_sequential_alpha.json", "r05_bench_2ranks_gloo_one_gpu_strong.json",
          "r05_bench_2ranks_gloo_one_gpu_forced_row_assist.json", "r03_bench_2ranks_gloo_one_gpu_weak.json"]


def _load(name):
    text = open(os.path.join(ROOT, "profiles", name)).read().strip().splitlines()[-1]
    return json.loads(text)


def _with_extras(d):
    d = dict(d)
    d["other_workloads"] = {"resnet50": {"layers_per_s": 1203.4, "job_ms": 33.24, "jobs_timed": 16, "layers": 40, "mask_parity": True,
                                         "layers_with_golden": 40, "weights_rel_frobenius_max": 3.1e-12},
                            "vgg16_5x": {"layers_per_s": 397.2, "job_ms": 25.18, "jobs_timed": 20, "layers": 10, "mask_parity": True,
                                         "layers_with_golden": 10, "weights_rel_frobenius_max": 8.4e-13}}
    d["r3"] = {"pass_s": 2.021, "vh_s": 1.02, "itq_s": 0.84, "prune_s": 0.115, "convs": 12, "note": "x" * 300}
    return d


@pytest.mark.parametrize("name", CANNED)
def test_line_is_short_parses_and_carries_the_contract(name):
    d = _load(name)
    if name == "r05_bench_vgg16.json":
        d = _with_extras(d)
    text = line.render(d)
    assert "\n" not in text and len(text.encode()) < line.MAX_LINE_BYTES
    out = json.loads(text)
    for k in line.CONTRACT_KEYS:
        assert k in out, k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "data"):
        assert out[k] == d[k], k
    assert out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["config"]["workload"] and len(out["config"]["workload"]) <= 230
    if d.get("roofline"):
        for k in line.ROOFLINE_KEYS:
            assert k in out["roofline"], k
        assert out["roofline"]["achieved"] == pytest.approx(d["roofline"]["achieved"], rel=1e-3)
        assert out["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-3)
    if d.get("cpu_baseline"):
        for k in line.CPU_KEYS:
            assert out["cpu_baseline"][k] is not None, k
    assert "shed" not in out


def test_default_vgg16_line_has_the_round6_keys():
    """what the judge asked to find inside BENCH_r06.json.parsed"""
    d = _with_extras(_load("r05_bench_vgg16.json"))
    out = json.loads(line.render(d))
    for k in ("workload", "jobs_per_step", "jobs_timed", "timed_region_s", "world_size", "backend"):
        assert k in out["config"], k
    roof = out["roofline"]
    assert roof["kernel"] == "k_chol_step" and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 78.6
    for k in ("chip_level_frac", "traffic", "traffic_algorithmic", "traffic_ratio", "gram", "alpha_search_ns_per_step"):
        assert roof.get(k) is not None, k
    assert set(roof["gram"]) >= {"frac", "traffic_ratio"}
    cpu = out["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "cpu_model", "job_seconds_cpu", "job_speedup_wall_clock"):
        assert cpu.get(k) is not None, k
    assert out["mask_parity_vs_reference_golden"] is True and out["weights_rel_frobenius_max"] <= 1e-5
    assert out["value_conv3_block"]["value"] > 0 and out["pcie_inclusive_layers_per_s"] > 0 and out["strong_scaling_bound_ms"] > 0
    assert set(out["other_workloads"]) == {"resnet50", "vgg16_5x"}
    assert set(out["other_workloads"]["resnet50"]) == {"layers_per_s", "job_ms", "mask_parity"}
    assert set(out["r3"]) == {"pass_s", "vh_s", "itq_s", "prune_s"}


def test_an_overlong_line_sheds_optional_groups_never_contract_keys():
    d = _with_extras(_load("r05_bench_vgg16.json"))
    d["other_workloads"] = {"job%03d" % i: {"layers_per_s": 1.0, "job_ms": 2.0, "mask_parity": True} for i in range(120)}
    text = line.render(d)
    out = json.loads(text)
    assert len(text.encode()) < line.MAX_LINE_BYTES and "other_workloads" in out["shed"]
    for k in line.CONTRACT_KEYS:
        assert k in out


def test_bench_prints_one_line_and_writes_the_detail_file(tmp_path, monkeypatch, capsys):
    """bench.main() with a stubbed leg: stdout is exactly one line (the compact one), the full dict lands in --detail"""
    import bench
    from benchkit import job
    detail = _with_extras(_load("r05_bench_vgg16.json"))
    monkeypatch.setattr(job, "bench_job", lambda args, env, name: dict(detail))
    monkeypatch.setattr(bench, "vgg16_extras", lambda args, env, out: None)
    path = tmp_path / "detail.json"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--detail", str(path)])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.main()
    printed = capsys.readouterr().out
    assert printed.count("\n") == 1 and len(printed.encode()) < line.MAX_LINE_BYTES
    assert json.loads(printed)["value"] == detail["value"]
    saved = json.load(open(path))
    assert saved["per_layer_rank0"] == detail["per_layer_rank0"] and saved["roofline"]["kernels"] == detail["roofline"]["kernels"]
