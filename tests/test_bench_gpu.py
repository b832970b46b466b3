"""bench.py's output contract on a real GPU: one JSON line with the driver's fields, the roofline and
cpu_baseline objects, and the per-config pipelines (hit counts equal to the pinned fixture's)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5",
                        "--cpu-seconds", "2", "--saturated-drones", "65536", "--config-reps", "10"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["rccl_ranks"] is None and d["backend"] is None      # one rank, no process group
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e5                                       # the north-star floor: 100 k trajectories/s
    assert abs(d["value"] - 256 * 20 / (d["ms_per_step"] * 20e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) < 1e-6
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > cb["single_thread_value"] * 0.5 and "sample" in cb
    assert cb["b1_one_lu_four_rhs"]["value"] > cb["value"]          # one LU for four right-hand sides is cheaper
    assert d["max_norm_rel_err_vs_oracle"] <= 1e-6
    cfg = d["configs"]
    assert set(cfg) == {"1", "2", "3", "4"}
    assert cfg["2"]["pairwise_hits"] == cfg["2"]["pairwise_hits_fixture"] > 0
    assert cfg["3"]["pairwise_hits"] == cfg["3"]["pairwise_hits_fixture"] > 0
    assert cfg["3"]["mesh_hits"] == cfg["3"]["mesh_hits_fixture"] > 0
    assert cfg["2"]["solve_failures"] == cfg["3"]["solve_failures"] == cfg["4"]["solve_failures"] == 0
    assert cfg["4"]["max_norm_rel_err_vs_oracle"] <= 1e-6
    assert set(cfg["3"]["stage_us"]) == {"solve", "sample", "pairwise", "mesh"}
    pw = cfg["2"]["stages"]["pairwise"]
    assert 0.0 < pw["frac"] < 1.0 and pw["speedup_vs_all_pairs_issue_peak"] >= pw["frac"] and "frac_on_all_pairs" not in pw
    bp = pw["broad_phase"]
    assert bp is not None and 0 < bp["shares_surviving"] <= bp["shares"] and 0 < bp["group_pairs_surviving"] <= bp["group_pairs"]
    assert 0.0 < bp["cull_ratio"] < 1.0
    assert cfg["2"]["stages"]["pairwise"]["pairs_evaluated_once_over_all_ranks"] is True
    assert cfg["4"]["stages"]["solve"]["kernel"] == "msnap::solve_kernel_twin<5, 10>"      # what the library says it launched
    assert cfg["4"]["max_norm_rel_err_vs_oracle"] <= 1e-9
    # counters are quoted only for the kernel sources they were taken from: a value comes with its source, a null
    # with the reason
    for rec in (d["roofline"], cfg["4"]["stages"]["solve"], d["saturated"]["roofline"]):
        assert isinstance(rec["traffic_source"], str) and len(rec["traffic_source"]) > 10
        assert rec["traffic"] is None or "csrc" in rec["traffic_source"]
    mesh = cfg["3"]["stages"]["mesh"]
    assert 0.0 < mesh["cull_ratio"] < 1.0 and 0.0 < mesh["frac"] < 1.0 and mesh["speedup_vs_all_pairs_issue_peak"] > mesh["frac"]
    ss = d["strong_scaling"]
    assert ss["n_gpus"] == 1 and {"solve_order9_65536x10", "solve_order7_65536x10", "formation_4096x10",
                                  "formation_16384x10"} <= set(ss)
    assert all(ss[k]["value"] > 1e6 for k in ss if isinstance(ss[k], dict))
    assert d["saturated"]["max_norm_rel_err_vs_oracle"] <= 1e-6
    assert d["shared_grid"]["saturated"]["max_norm_rel_err_vs_oracle"] <= 1e-6
    assert d["end_to_end"]["value"] > 0
