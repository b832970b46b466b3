"""Sharding a swarm over the GPUs of one node (one process per GPU).

The solve needs no communication: drones are split into contiguous blocks, one
per rank.  The formation (drone-vs-drone) pass has one real exchange step: every
rank samples its own shard, the sampled positions are all-gathered (RCCL over
xGMI when the process group is "nccl"; "gloo" on CPU for the tests) and each
rank then checks its own rows against all columns -- row-owned results, no
reduction.  The mesh sweep replicates the (tiny) mesh and shards the drones.

torch / torch.distributed are plumbing here (device memory + the collective);
all arithmetic happens in libmsnap through the `compute` object, by default a
`DeviceCompute` around a Context.  The CPU tests plug in a stand-in with the
same three methods to exercise the exchange logic under gloo.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> tuple:
    """Contiguous block partition: the first n % world ranks get one extra drone."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_sizes(n: int, world: int) -> list:
    return [shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)]


class DeviceCompute:
    """The three device operations of the formation pipeline, on torch CUDA tensors."""

    def __init__(self, ctx, torch_module):
        self.ctx = ctx
        self.torch = torch_module
        self.device = torch_module.device("cuda", ctx.device_id)
        # run on torch's current stream so the collective and the kernels order naturally
        ctx.set_stream(torch_module.cuda.current_stream(self.device).cuda_stream)

    def solve(self, wp, t):
        torch = self.torch
        n, m, _ = wp.shape
        M = m - 1
        coef = torch.empty((n, M, 4, self.ctx.ncoef), dtype=torch.float64, device=self.device)
        dur = torch.empty((n, M), dtype=torch.float64, device=self.device)
        status = torch.empty((n,), dtype=torch.int32, device=self.device)
        if n:
            self.ctx.solve_batch_device(n, M, wp, t, t.dim() == 1, coef, dur, status)
        return coef, dur, status

    def sample(self, coef, dur, dt, n_samples):
        torch = self.torch
        n, M = dur.shape
        pos = torch.empty((n, n_samples, 3), dtype=torch.float64, device=self.device)
        if n:
            self.ctx.sample_device(n, M, coef, dur, dt, n_samples, 3, pos)
        return pos

    def collide(self, pos_rows, row_offset, pos_all, radius):
        torch = self.torch
        r = pos_rows.shape[0]
        md = torch.empty((r,), dtype=torch.float64, device=self.device)
        partner = torch.empty((r,), dtype=torch.int32, device=self.device)
        hit = torch.empty((r,), dtype=torch.int32, device=self.device)
        if r:
            self.ctx.formation_collide_device(r, row_offset, pos_all.shape[0], pos_rows.shape[1], pos_rows,
                                              pos_all, radius, md, partner, hit)
        return md, partner, hit

    def mesh(self, pos, tris, radius):
        torch = self.torch
        n = pos.shape[0]
        md = torch.empty((n,), dtype=torch.float64, device=self.device)
        hit = torch.empty((n,), dtype=torch.int32, device=self.device)
        if n:
            self.ctx.mesh_sweep_device(n, pos.shape[1], pos, tris.shape[0], tris, radius, md, hit)
        return md, hit


@dataclass
class FormationResult:
    lo: int              # first global drone index owned by this rank
    hi: int
    min_dist: object     # [hi-lo]
    partner: object      # [hi-lo] global index
    hit: object          # [hi-lo]
    positions_all: object  # [N, S, 3] after the all-gather


def all_gather_positions(pos_local, n_total: int, world: int, rank: int, dist, torch):
    """All-gather the ranks' [n_r, S, 3] position blocks into [N, S, 3].

    Shards may differ by one drone, so every rank pads to the largest shard, one
    `all_gather_into_tensor` moves the padded blocks (a single collective: RCCL
    picks a direct all-gather on the xGMI full mesh; the message is latency bound,
    SURVEY.md 8e) and the padding is dropped afterwards."""
    sizes = shard_sizes(n_total, world)
    if world == 1:
        return pos_local
    S = pos_local.shape[1]
    nmax = max(sizes)
    gathered = torch.empty((world * nmax, S, 3), dtype=pos_local.dtype, device=pos_local.device)
    if all(s == nmax for s in sizes):
        # even shards (4096 drones on 2, 4 or 8 GPUs): the shard itself is the send buffer
        dist.all_gather_into_tensor(gathered, pos_local.contiguous())
        return gathered
    padded = torch.zeros((nmax, S, 3), dtype=pos_local.dtype, device=pos_local.device)
    padded[:pos_local.shape[0]] = pos_local
    dist.all_gather_into_tensor(gathered, padded)
    parts = [gathered[r * nmax:r * nmax + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def formation_pass(compute, coef_local, dur_local, n_total: int, world: int, rank: int, dt: float,
                   n_samples: int, radius: float, dist=None, torch=None, status_local=None) -> FormationResult:
    """Sample the local shard, exchange, collide own rows against everybody.

    `status_local` (the solve's per-drone status of this shard): a failed solve leaves NaN
    coefficients, and NaN samples never win a minimum (include/msnap.h) -- such a drone would
    read as collision-free and be invisible to the others, so it is refused here."""
    lo, hi = shard_bounds(n_total, world, rank)
    if status_local is not None and int(abs(status_local).sum()) != 0:
        raise ValueError("formation_pass: the solve reported failed drones (status != 0) in rows "
                         f"[{lo}, {hi}); their samples are NaN and cannot be collision-checked")
    pos_local = compute.sample(coef_local, dur_local, dt, n_samples)
    pos_all = all_gather_positions(pos_local, n_total, world, rank, dist, torch) if world > 1 else pos_local
    md, partner, hit = compute.collide(pos_local, lo, pos_all, radius)
    return FormationResult(lo, hi, md, partner, hit, pos_all)


def default_sample_count(total_duration: float, dt: float) -> int:
    """len(np.arange(0, duration, dt)) -- the reference's sampling loop
    (src/trajectory_visualising/visualization.py:53)."""
    return int(len(np.arange(0.0, total_duration, dt)))
