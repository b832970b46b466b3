"""The RCCL branch of the formation pass, executed on the one GPU of the box: a world-size-1 `nccl` process group and
the SAME `all_gather_into_tensor` calls `swarm.all_gather_positions` / `all_gather_parts` make on several GPUs
(`force=True`: issued on a one-rank group too), on device tensors, ordered against the context's borrowed stream
(torch's current stream), followed by the part / fold kernels or the whole-swarm pass -- equal to the no-collective
path and to the oracle.  What this cannot show is a second rank: the multi-GPU run itself is the driver's."""
import os
import socket

import numpy as np
import pytest

import c_oracle
from drone_path_planning_python_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_world1():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", device_id=dev)      # RCCL
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    yield dist
    dist.barrier()
    dist.destroy_process_group()
    for k in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)


def _swarm(n, M, seed):
    wp, t = synthetic.swarm(seed, n, M)
    wp[..., :3] *= 0.35          # crowd it: some pairs collide
    return wp, t


@pytest.mark.parametrize("n,mode", [(1500, "parts"), (3072, "parts"), (3072, "whole")])
def test_formation_pass_through_rccl_on_one_rank(nccl_world1, ctx7, n, mode):
    import torch
    from drone_path_planning_python_amd import swarm as sw
    dist = nccl_world1
    dev = torch.device("cuda", 0)
    wp, t = _swarm(n, 6, 17 + n)
    S, dt = 31, 0.25
    comp = sw.DeviceCompute(ctx7, torch)        # borrows torch's current stream: the collectives order against it
    try:
        coef, dur, status = comp.solve(torch.from_numpy(wp).to(dev), torch.from_numpy(t).to(dev))
        assert int(status.abs().sum()) == 0
        md0 = sw.formation_pass(comp, coef, dur, n, 1, 0, dt, S, 0.0, torch=torch, status_local=status).min_dist
        radius = 0.5 * float(md0.median()) * 1.0001      # about half of the swarm collides
        plain = sw.formation_pass(comp, coef, dur, n, 1, 0, dt, S, radius, torch=torch, status_local=status)
        coll = sw.formation_pass(comp, coef, dur, n, 1, 0, dt, S, radius, dist=dist, torch=torch, status_local=status,
                                 force_collectives=True, force_mode=mode)
        torch.cuda.synchronize()
        # the gathered block is a new tensor (the collective ran), with the shard's content
        assert coll.positions_all.data_ptr() != plain.positions_all.data_ptr()
        assert torch.equal(coll.positions_all, plain.positions_all)
        for a, b in ((coll.min_dist, plain.min_dist), (coll.partner, plain.partner), (coll.hit, plain.hit)):
            assert torch.equal(a, b)
        md, partner, hit = c_oracle.formation_collide(plain.positions_all.cpu().numpy(), radius)
        np.testing.assert_array_equal(coll.min_dist.cpu().numpy(), md)
        np.testing.assert_array_equal(coll.partner.cpu().numpy(), partner)
        np.testing.assert_array_equal(coll.hit.cpu().numpy().astype(bool), hit)
        assert 0 < hit.sum() < n
        if mode == "whole":
            assert ctx7.get_option("collide_last_cull") == 1
    finally:
        comp.close()
        ctx7.use_own_stream()


def test_the_two_collectives_themselves(nccl_world1, ctx7):
    """all_gather_positions / all_gather_parts with force=True on device tensors: one all_gather_into_tensor each,
    results equal to their inputs on a one-rank group; uneven shard sizes cannot occur at world 1, the padded branch is
    covered under gloo (tests/test_distributed_cpu.py)."""
    import torch
    from drone_path_planning_python_amd import swarm as sw
    dist = nccl_world1
    dev = torch.device("cuda", 0)
    pos = torch.randn((257, 13, 3), dtype=torch.float64, device=dev)
    out = sw.all_gather_positions(pos, 257, 1, 0, dist, torch, force=True)
    part = torch.randint(0, 255, (ctx7.formation_part_bytes(257),), dtype=torch.uint8, device=dev)
    parts = sw.all_gather_parts(part, 1, dist, torch, force=True)
    torch.cuda.synchronize()
    assert out.data_ptr() != pos.data_ptr() and torch.equal(out, pos)
    assert parts.shape == (1, part.shape[0]) and torch.equal(parts[0], part)
    # the decision all-reduce of note_whole_pass on the nccl backend (device tensor, MIN)
    flag = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    assert int(flag.item()) == 1


def test_bench_one_rank_process_group():
    """`python bench.py --force-pg`: the bench's own nccl branch (init_process_group(device_id=...), barriers, the
    max / sum all-reduces on device tensors) on a one-rank group."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-pg", "--steps", "10", "--warmup", "3",
                        "--config-reps", "3", "--no-saturated", "--no-shared-grid", "--no-strong", "--no-cpu-baseline",
                        "--no-end-to-end"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["backend"] == "nccl (RCCL)"
    assert d["configs"]["2"]["pairwise_hits"] == d["configs"]["2"]["pairwise_hits_fixture"]
