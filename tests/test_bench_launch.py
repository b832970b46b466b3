"""bench.py's rank count: `--gpus N` and the launcher environment must agree, and a plain `python bench.py --gpus N`
starts its N ranks itself (before it touches HIP).  The CPU half uses `--launch-check` (process group + one all-reduce,
no kernels) over gloo; the GPU half runs the whole bench as 2 gloo ranks sharing the box's one GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _json_lines(out):
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_plain_gpus_2_starts_two_ranks():
    """no torchrun on the command line, no launcher environment: bench.py is the launcher"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["ranks_counted_by_all_reduce"] == 2 and d["backend"] == "gloo"


def test_plain_gpus_8_starts_eight_ranks():
    """the driver's largest case as far as a CPU box can take it: eight ranks, one all-reduce over all of them"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--backend", "gloo", "--launch-check"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_lines(r.stdout)[0]
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["ranks_counted_by_all_reduce"] == 8


def test_under_torchrun_the_flag_must_match_world_size():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
            "--master-port", "29631", BENCH, "--backend", "gloo", "--launch-check"]
    ok = subprocess.run(base + ["--gpus", "2"], capture_output=True, text=True, timeout=300, env=_env())
    assert ok.returncode == 0, ok.stderr[-2000:]
    assert _json_lines(ok.stdout)[0]["n_gpus"] == 2
    bad = subprocess.run(base + ["--gpus", "4"], capture_output=True, text=True, timeout=300, env=_env())
    assert bad.returncode != 0
    assert not _json_lines(bad.stdout) and "WORLD_SIZE=2" in bad.stderr


def test_gpus_flag_against_a_stale_world_size_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120,
                       env=_env(WORLD_SIZE="1"))
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "--gpus 2" in r.stderr and "WORLD_SIZE=1" in r.stderr


def test_more_ranks_than_gpus_over_rccl_is_refused():
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode != 0 and not _json_lines(r.stdout)
    assert "needs 8 visible GPUs" in r.stderr


@pytest.mark.gpu
def test_plain_gpus_2_runs_the_sharded_bench_on_the_box():
    """the whole bench as `python bench.py --gpus 2` (gloo rehearsal: both ranks on this box's one GPU, the collectives
    on host copies): n_gpus == 2 and the sharded formation configs reproduce the fixture's hit counts"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "10", "--warmup", "3",
                        "--config-reps", "3", "--no-saturated", "--no-shared-grid", "--no-strong"],
                       capture_output=True, text=True, timeout=900, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert d["cpu_baseline"] is None          # (reported at N = 1 only)
    cfg = d["configs"]
    assert cfg["2"]["rccl_ranks"] == 2
    assert cfg["2"]["pairwise_hits"] == cfg["2"]["pairwise_hits_fixture"] > 0
    assert cfg["3"]["pairwise_hits"] == cfg["3"]["pairwise_hits_fixture"] > 0
    assert cfg["3"]["mesh_hits"] == cfg["3"]["mesh_hits_fixture"] > 0
    assert cfg["4"]["solve_failures"] == 0 and cfg["4"]["max_norm_rel_err_vs_oracle"] <= 1e-9
