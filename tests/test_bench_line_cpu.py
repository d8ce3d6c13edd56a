"""CPU: bench.py's final stdout line is the one the driver parses -- under bench.LINE_LIMIT bytes, valid JSON, carrying the
contract's keys, `roofline` and `cpu_baseline` -- whatever the run's full result looks like (round 5's grew to 22.5 KB and
was not parsed).  The canned result is a full result of a real default run (profiles/r05_z_bench_default_line.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_z_bench_default_line.json")))


def test_line_is_short_and_round_trips():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 20000            # the result that did not parse
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 6144
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["metric"] == full["metric"] and d["unit"] == "read-pairs/s" and d["n_gpus"] == 1
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5 and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-4
    assert len(d["config"]["workload"]) <= 200 and d["config"]["pairs_per_gpu"] == 10_000_000
    rf = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_alone", "frac_step", "traffic", "algorithmic_bytes_per_launch",
              "avg_kernel_ms", "avg_kernel_ms_alone", "launches", "traffic_measured_in_run"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["bound"] == "hbm"
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and len(cb["sample"]) <= 160
    assert cb["all_cores"]["value"] > 0 and cb["files_to_files"]["value"] > 0 and cb["reference_calibration"]["value"] == 11000.0
    me = d["metric_e2e"]
    assert me["value"] > 0 and me["large"]["value"] > 0 and me["marginal_pairs_per_s"] > 0 and me["fixed_s"] > 0
    assert me["grch38"]["cold"] > 0 and me["grch38"]["warm"] > 0 and me["checked_against_oracle"] is True and me["cpu_files_identical"] is True
    assert "kernels" not in d and "e2e" not in d


def test_line_survives_odd_results():
    import bench
    full = _canned()
    # a rank>0-less multi-GPU result, a failed e2e leg, prose of any length, no CPU baseline
    full["cpu_baseline"] = None
    full["e2e"] = {"error": "x" * 5000, "ok": False}
    full["metric_e2e"] = {"value": None, "unit": "u", "checked_against_oracle": False}
    full["config"]["workload"] = "w" * 3000
    full["exchange"] = {"transport": "rccl", "n_ranks": 8, "ranks": 8, "calls": 12, "bytes_per_rank": 1 << 20, "us_per_step": 65.0, "junk": "j" * 9000}
    full["per_rank_ms_per_step"] = [5.4] * 8
    line = bench.compact_line(full)
    d = json.loads(line)
    assert len(line) < bench.LINE_LIMIT and d["cpu_baseline"] is None and len(d["metric_e2e"]["error"]) <= 300
    assert d["exchange"]["transport"] == "rccl" and "junk" not in d["exchange"] and len(d["per_rank_ms_per_step"]) == 8


def test_detail_file(tmp_path):
    import bench
    full = _canned()
    p = str(tmp_path / "bench_detail.json")
    w = bench.write_detail(full, p)
    assert p in w and len(json.load(open(p))["kernels"]) == 16
