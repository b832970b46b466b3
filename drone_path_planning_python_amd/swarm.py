"""Sharding a swarm over the GPUs of one node (one process per GPU).

The solve needs no communication: drones are split into contiguous blocks, one
per rank.  The formation (drone-vs-drone) pass exchanges twice: every rank samples
its own shard and the sampled positions are all-gathered (RCCL over xGMI when the
process group is "nccl"; "gloo" on CPU for the tests); then the swarm's unordered
pairs -- one triangular line of (row block, column) units -- are split into `world`
equal contiguous parts, rank r evaluates part r (every pair on exactly ONE rank,
both drones credited), the per-drone partial minima (12 bytes per drone and rank)
are all-gathered and every rank folds them for the rows it owns.  The mesh sweep
replicates the (tiny) mesh and shards the drones; it depends on the rank's own
samples only, so with a second context (`side_ctx`) it runs on a side stream next
to the exchanges and the pairwise pass.

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
    """The device operations of the formation pipeline, on torch CUDA tensors.

    `side_ctx`: a second Context on the same device.  When given, `mesh_begin` launches the mesh sweep
    on a side stream (ordered after what the main stream holds at that moment) and `mesh_end` makes the
    main stream wait for it: the sweep overlaps the all-gather and the pairwise pass, whose launch ends
    with SIMDs running out of shares."""

    _reuse = None      # (class default: no output reuse)

    def __init__(self, ctx, torch_module, side_ctx=None, reuse_outputs=False):
        """`reuse_outputs`: every method hands back the SAME output tensors on each call with the same shapes (valid
        until its next call) instead of allocating new ones -- for a loop that consumes a pass's results before it
        starts the next: eleven `torch.empty` per formation pipeline are ~30 us of host time, as much as the host has
        to spare beside 80 us of kernels."""
        self.ctx = ctx
        self.torch = torch_module
        self._reuse = {} if reuse_outputs else None
        self.device = torch_module.device("cuda", ctx.device_id)
        # run on torch's current stream so the collective and the kernels order naturally
        self.main = torch_module.cuda.current_stream(self.device)
        ctx.set_stream(self.main.cuda_stream)
        self.side_ctx = side_ctx
        self.side = None
        if side_ctx is not None:
            if side_ctx.device_id != ctx.device_id:
                raise ValueError("side_ctx must live on the same device")
            # the side context's own stream at the lowest priority: the sweep's workgroups are dispatched
            # when the main stream's kernels have none waiting, i.e. into the end of the pairwise launch
            side_ctx.set_option("own_stream_priority", 1)
            # ... and as a grid of 12 waves per CU that walks the drones: one workgroup per drone holds every wave slot
            # of the chip for the sweep's whole run, and the main stream's next kernel waits for its end
            if side_ctx.get_option("mesh_waves_per_cu") == 0:
                side_ctx.set_option("mesh_waves_per_cu", 12)
            side_ctx.use_own_stream()
            self.side = torch_module.cuda.ExternalStream(side_ctx.stream(), device=self.device)
        self._mesh_pending = None
        self._whole_ok = {}      # (n_total, n_samples, world) -> {"ok": the whole-swarm pass pays, "since": passes since decided}

    def _out(self, name, shape, dtype):
        if self._reuse is None:
            return self.torch.empty(shape, dtype=dtype, device=self.device)
        key = (name, tuple(shape), dtype)
        buf = self._reuse.get(key)
        if buf is None:
            buf = self._reuse[key] = self.torch.empty(shape, dtype=dtype, device=self.device)
        return buf

    def solve(self, wp, t):
        torch = self.torch
        n, m, _ = wp.shape
        M = m - 1
        coef = self._out("solve.coef", (n, M, 4, self.ctx.ncoef), torch.float64)
        dur = self._out("solve.dur", (n, M), torch.float64)
        status = self._out("solve.status", (n,), torch.int32)
        if n:
            self.ctx.solve_batch_device(n, M, wp, t, t.dim() == 1, coef, dur, status)
        return coef, dur, status

    def solve_grid(self, wp):
        """The same solve on the shared time grid the context was prepared for (`Context.prepare_grid`): one
        fp64 MFMA GEMM against the grid's operator instead of the per-drone recurrence."""
        torch = self.torch
        n, m, _ = wp.shape
        M = m - 1
        # (the library refuses a segment count that is not the prepared grid's; the tensor's shape is checked here)
        if wp.dim() != 3 or wp.shape[2] != 4 or m != self.ctx.grid_waypoints():
            raise ValueError(f"solve_grid: wp must be [n, {self.ctx.grid_waypoints()}, 4] for the prepared grid, got {tuple(wp.shape)}")
        coef = self._out("solve_grid.coef", (n, M, 4, self.ctx.ncoef), torch.float64)
        dur = self._out("solve_grid.dur", (n, M), torch.float64)
        status = self._out("solve_grid.status", (n,), torch.int32)
        if n:
            self.ctx.solve_grid_device(n, M, wp, coef, dur, status)
        return coef, dur, status

    def sample(self, coef, dur, dt, n_samples):
        torch = self.torch
        n, M = dur.shape
        pos = self._out("sample.pos", (n, n_samples, 3), torch.float64)
        if n:
            self.ctx.sample_device(n, M, coef, dur, dt, n_samples, 3, pos)
        return pos

    def sample_rows_t(self, coef, dur, dt, n_samples, n_cols=None):
        """(pos [n, S, 3], pos_t): the sampler's second output is what the pairwise pass over these drones would
        otherwise compute in a launch of its own -- the transposed row image [S][3][pitch], or (a whole swarm behind the
        exact broad phase) the drones' path boxes and sort keys; `collide(..., rows_t=pos_t)` hands it over.  With
        `n_cols` (the columns these rows, as a shard at offset 0, will meet) the library is asked first whether that
        pass reads a hand-over at all (paths shorter than 6 samples do not), and pos_t is None if not."""
        torch = self.torch
        n, M = dur.shape
        if n_cols is not None and not self.ctx.collide_reads_rows_t(n, 0, n_cols, n_samples):
            return self.sample(coef, dur, dt, n_samples), None
        pos = self._out("sample_rows_t.pos", (n, n_samples, 3), torch.float64)
        pos_t = self._out("sample_rows_t.pos_t", (self.ctx.collide_rows_t_doubles(n, n_samples),), torch.float64)
        if n:
            self.ctx.sample_collide_device(n, M, coef, dur, dt, n_samples, pos, pos_t)
        return pos, pos_t

    def solve_grid_sample(self, wp, dt, n_samples, n_cols=None):
        """solve_grid and the sampler as one launch (msnap_solve_grid_sample_device): (coef, dur, status, pos, pos_t),
        the same tensors, bit for bit, as `solve_grid` followed by `sample_rows_t` (pos_t is None where that would
        return None, or when `n_cols` is None and no hand-over is wanted)."""
        torch = self.torch
        n, m, _ = wp.shape
        M = m - 1
        if wp.dim() != 3 or wp.shape[2] != 4 or m != self.ctx.grid_waypoints():
            raise ValueError(f"solve_grid_sample: wp must be [n, {self.ctx.grid_waypoints()}, 4] for the prepared grid, got {tuple(wp.shape)}")
        coef = self._out("solve_grid_sample.coef", (n, M, 4, self.ctx.ncoef), torch.float64)
        dur = self._out("solve_grid_sample.dur", (n, M), torch.float64)
        status = self._out("solve_grid_sample.status", (n,), torch.int32)
        pos = self._out("solve_grid_sample.pos", (n, n_samples, 3), torch.float64)
        pos_t = None
        if n_cols is not None and self.ctx.collide_reads_rows_t(n, 0, n_cols, n_samples):
            pos_t = self._out("solve_grid_sample.pos_t", (self.ctx.collide_rows_t_doubles(n, n_samples),), torch.float64)
        if n:
            self.ctx.solve_grid_sample_device(n, M, wp, dt, n_samples, coef, dur, status, pos, pos_t)
        return coef, dur, status, pos, pos_t

    def collide(self, pos_rows, row_offset, pos_all, radius, rows_t=None):
        torch = self.torch
        r = pos_rows.shape[0]
        md = self._out("collide.md", (r,), torch.float64)
        partner = self._out("collide.partner", (r,), torch.int32)
        hit = self._out("collide.hit", (r,), torch.int32)
        if r and rows_t is not None:
            self.ctx.formation_collide_t_device(r, row_offset, pos_all.shape[0], pos_rows.shape[1], rows_t, pos_rows,
                                                pos_all, radius, md, partner, hit)
        elif r:
            self.ctx.formation_collide_device(r, row_offset, pos_all.shape[0], pos_rows.shape[1], pos_rows,
                                              pos_all, radius, md, partner, hit)
        return md, partner, hit

    # ---- several ranks: who evaluates which pairs -------------------------------------------------------------
    # "parts": every unordered pair on exactly one rank (collide_part), a second small all-gather, the fold -- the
    # arithmetic is divided by the number of ranks.  "whole": every rank runs the pass over the WHOLE gathered swarm
    # behind the exact broad phase and keeps its rows -- no second collective, and on swarms where the broad phase culls
    # (4096-drone formation fixture: 3 % of the pairs are evaluated, 68 us) that is less than a rank's part of all pairs
    # plus its collective.  Both inputs of the choice are the library's: whether a whole pass of this shape would run
    # behind the broad phase (msnap_formation_collide_takes_broad_phase) and whether, by the counts a whole pass left,
    # it pays against the parts (msnap_formation_whole_pass_pays) -- no threshold or cost constant lives here.  Every
    # rank ran the pass on the same positions and gets the same counts; they still agree through one MIN all-reduce.
    REPROBE_EVERY = 256      # passes after which a shape that went to the parts tries one whole pass again

    def pairwise_mode(self, n_total, n_samples, world):
        key = (int(n_total), int(n_samples), int(world))
        if world == 1:
            return "parts"
        st = self._whole_ok.get(key)
        if st is not None and not st["ok"] and st["since"] < self.REPROBE_EVERY:
            return "parts"
        return "whole" if self.ctx.collide_takes_broad_phase(n_total, 0, n_total, n_samples) else "parts"

    def note_parts_pass(self, n_total, n_samples, world):
        """After a pass in parts: counts towards the next re-probe of the whole-swarm mode (a swarm that was dense
        when the decision was taken may have spread out since)."""
        st = self._whole_ok.get((int(n_total), int(n_samples), int(world)))
        if st is not None and not st["ok"]:
            st["since"] += 1

    def note_whole_pass(self, n_total, n_samples, world, dist=None):
        """After a whole-swarm pass: decide whether it pays against the parts -- on the first pass of a swarm shape,
        on a re-probe, and every REPROBE_EVERY whole passes (a sparse swarm may have contracted).  Reads the pass's
        survivor counts (one stream synchronisation).  With `dist` the ranks take the decision together (a MIN
        all-reduce of one flag): a split decision would leave them in different collectives, so the all-reduce is
        entered whatever happened locally -- a rank whose query failed contributes 0 (parts) and re-raises after it."""
        key = (int(n_total), int(n_samples), int(world))
        st = self._whole_ok.get(key)
        if st is not None and st["ok"]:
            st["since"] += 1
            if st["since"] < self.REPROBE_EVERY:
                return
        err = None
        try:
            ok = bool(self.ctx.whole_pass_pays(n_total, world))
        except Exception as e:      # (e.g. MSNAP_ECAPTURE: the query synchronises, the stream is being captured)
            ok, err = False, e
        if dist is not None and world > 1:
            flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32,
                                     device=self.device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        self._whole_ok[key] = {"ok": ok, "since": 0}
        if err is not None:
            raise err

    def collide_part(self, pos_all, part, n_parts):
        """This rank's part of the pass over the whole swarm: uint8 [formation_part_bytes(N)] (squared minima
        of every drone over the pairs of this part, then the partners)."""
        n = pos_all.shape[0]
        out = self._out("collide_part.out", (self.ctx.formation_part_bytes(n),), self.torch.uint8)
        if n:
            self.ctx.formation_collide_part_device(n, pos_all.shape[1], pos_all, part, n_parts, out)
        return out

    def collide_finish(self, parts, n_total, row_offset, n_rows, radius):
        """Fold the gathered parts (uint8 [P, formation_part_bytes(N)]) for the rows this rank owns."""
        torch = self.torch
        md = self._out("collide_finish.md", (n_rows,), torch.float64)
        partner = self._out("collide_finish.partner", (n_rows,), torch.int32)
        hit = self._out("collide_finish.hit", (n_rows,), torch.int32)
        if n_rows:
            self.ctx.formation_collide_finish_device(n_total, parts.shape[0], parts, row_offset, n_rows, radius,
                                                     md, partner, hit)
        return md, partner, hit

    def _mesh_on(self, ctx, pos, tris, radius, md=None, hit=None):
        torch = self.torch
        n = pos.shape[0]
        if md is None:
            md = self._out("mesh.md", (n,), torch.float64)
            hit = self._out("mesh.hit", (n,), torch.int32)
        if n:
            ctx.mesh_sweep_device(n, pos.shape[1], pos, tris.shape[0], tris, radius, md, hit)
        return md, hit

    def mesh(self, pos, tris, radius):
        return self._mesh_on(self.ctx, pos, tris, radius)

    def mesh_begin(self, pos, tris, radius):
        """Start the mesh sweep; with a side context it runs next to whatever the main stream does until
        `mesh_end`, otherwise it is an ordinary launch on the main stream.

        All buffers are main-stream allocations: the side stream starts after everything the main stream
        holds now (so a recycled output block is quiet), the inputs are kept referenced until `mesh_end`,
        and `mesh_end` orders the main stream behind the sweep before anything can be recycled."""
        if self._mesh_pending is not None:
            raise RuntimeError("mesh_begin: a sweep is already pending")
        if self.side is None:
            self._mesh_pending = (self.mesh(pos, tris, radius), None)
            return
        n = pos.shape[0]
        md = self._out("mesh.md", (n,), self.torch.float64)
        hit = self._out("mesh.hit", (n,), self.torch.int32)
        self.side.wait_stream(self.main)            # the samples (and the mesh) are ready
        self._mesh_pending = (self._mesh_on(self.side_ctx, pos, tris, radius, md, hit), (pos, tris))

    def mesh_end(self):
        """(min_dist, hit) of the pending sweep, ordered into the main stream."""
        if self._mesh_pending is None:
            raise RuntimeError("mesh_end without mesh_begin")
        (out, _keep), self._mesh_pending = self._mesh_pending, None
        if self.side is not None:
            self.main.wait_stream(self.side)
        return out

    def mesh_abort(self):
        """Join a pending sweep and forget it: what a caller's `finally` runs when something between
        `mesh_begin` and `mesh_end` raised, so that the main stream is ordered behind the side stream before
        the sweep's buffers can be recycled and the next `mesh_begin` is not refused."""
        if self._mesh_pending is not None:
            self.mesh_end()

    def close(self):
        """Drop the wrapper of the side context's stream (call before closing `side_ctx`)."""
        self.mesh_abort()
        self.side = None


@dataclass
class FormationResult:
    lo: int              # first global drone index owned by this rank
    hi: int
    min_dist: object     # [hi-lo]
    partner: object      # [hi-lo] global index
    hit: object          # [hi-lo]
    positions_all: object  # [N, S, 3] after the all-gather
    mesh_min_dist: object = None   # [hi-lo] when a mesh was given
    mesh_hit: object = None


def all_gather_positions(pos_local, n_total: int, world: int, rank: int, dist, torch, force: bool = False):
    """All-gather the ranks' [n_r, S, 3] position blocks into [N, S, 3].  `force`: issue the collective on a
    one-rank group too (a one-GPU box then runs the same RCCL call a multi-GPU job makes).

    Shards may differ by one drone, so every rank pads to the largest shard, one
    `all_gather_into_tensor` moves the padded blocks (a single collective: RCCL
    picks a direct all-gather on the xGMI full mesh; the message is latency bound,
    SURVEY.md 8e) and the padding is dropped afterwards."""
    sizes = shard_sizes(n_total, world)
    if world == 1 and not force:
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


def all_gather_parts(part_local, world: int, dist, torch, force: bool = False):
    """All-gather the ranks' part blocks (uint8 [B] each) into [world, B]: the second, small collective of the
    formation pass (12 bytes per drone and rank).  `force` as in all_gather_positions."""
    if world == 1 and not force:
        return part_local.reshape(1, -1)
    gathered = torch.empty((world * part_local.shape[0],), dtype=part_local.dtype, device=part_local.device)
    dist.all_gather_into_tensor(gathered, part_local.contiguous())
    return gathered.reshape(world, -1)


def formation_pass(compute, coef_local, dur_local, n_total: int, world: int, rank: int, dt: float,
                   n_samples: int, radius: float, dist=None, torch=None, status_local=None,
                   mesh_tris=None, force_collectives: bool = False, force_mode: str = None,
                   _sampled=None) -> FormationResult:
    """Sample the local shard, exchange, evaluate this rank's part of the swarm's pairs, exchange the
    partial minima and fold them for the own rows (one rank: one symmetric launch, no exchange); with
    `mesh_tris` ([T, 3, 3]) also sweep the local shard against the mesh, started right behind the sampler
    so that a compute object with a side stream overlaps it with the exchanges and the pairwise pass.

    `status_local` (the solve's per-drone status of this shard): a failed solve leaves NaN
    coefficients, and NaN samples never win a minimum (include/msnap.h) -- such a drone would
    read as collision-free and be invisible to the others, so it is refused here.

    `force_collectives`: take the several-rank route -- both collectives, parts and fold, or the whole-swarm mode --
    on a one-rank group as well (what a one-GPU box can execute of the multi-GPU path); `force_mode` ("parts" /
    "whole") pins the pairwise mode instead of asking `compute.pairwise_mode`."""
    lo, hi = shard_bounds(n_total, world, rank)
    if status_local is not None and int(abs(status_local).sum()) != 0:
        raise ValueError("formation_pass: the solve reported failed drones (status != 0) in rows "
                         f"[{lo}, {hi}); their samples are NaN and cannot be collision-checked")
    multi = world > 1 or force_collectives
    if multi:
        mode = force_mode or (compute.pairwise_mode(n_total, n_samples, world) if hasattr(compute, "pairwise_mode") else "parts")
        # checked before the first collective: a compute object without the calls this mode needs must not leave
        # the other ranks waiting in an all-gather this rank never enters
        if mode == "parts" and not (hasattr(compute, "collide_part") and hasattr(compute, "collide_finish")):
            raise TypeError("formation_pass on several ranks needs compute.collide_part / collide_finish "
                            "(every pair on exactly one rank); this compute object only has collide()")
    rows_t = None
    if _sampled is not None:      # (formation_pass_from_waypoints: solve and sampler were one launch)
        pos_local, rows_t = _sampled
    elif not multi and hasattr(compute, "sample_rows_t"):
        pos_local, rows_t = compute.sample_rows_t(coef_local, dur_local, dt, n_samples, n_cols=n_total)
    else:
        pos_local = compute.sample(coef_local, dur_local, dt, n_samples)
    overlapped = mesh_tris is not None and hasattr(compute, "mesh_begin")
    if overlapped:
        compute.mesh_begin(pos_local, mesh_tris, radius)
    try:
        if multi:
            pos_all = all_gather_positions(pos_local, n_total, world, rank, dist, torch, force=force_collectives)
            if mode == "whole":
                md, partner, hit = (x[lo:hi] for x in compute.collide(pos_all, 0, pos_all, radius))
                if hasattr(compute, "note_whole_pass"):
                    compute.note_whole_pass(n_total, n_samples, world, dist if hasattr(dist, "get_backend") else None)
            else:
                part = compute.collide_part(pos_all, rank, world)
                parts = all_gather_parts(part, world, dist, torch, force=force_collectives)
                md, partner, hit = compute.collide_finish(parts, n_total, lo, hi - lo, radius)
                if hasattr(compute, "note_parts_pass"):
                    compute.note_parts_pass(n_total, n_samples, world)
        else:
            pos_all = pos_local
            if rows_t is not None:
                md, partner, hit = compute.collide(pos_local, lo, pos_all, radius, rows_t=rows_t)
            else:
                md, partner, hit = compute.collide(pos_local, lo, pos_all, radius)
        mmd = mhit = None
        if overlapped:
            mmd, mhit = compute.mesh_end()
        elif mesh_tris is not None:
            mmd, mhit = compute.mesh(pos_local, mesh_tris, radius)
    except BaseException as first:
        # join the side stream before the sweep's buffers can be recycled -- and keep the FIRST exception: after a
        # HIP error the join will most likely fail too and would otherwise replace the cause
        if overlapped and hasattr(compute, "mesh_abort"):
            try:
                compute.mesh_abort()
            except Exception as second:
                first.__context__ = second
        raise
    return FormationResult(lo, hi, md, partner, hit, pos_all, mmd, mhit)


def formation_pass_from_waypoints(compute, wp_local, n_total: int, world: int, rank: int, dt: float, n_samples: int,
                                  radius: float, dist=None, torch=None, mesh_tris=None, check_status: bool = True,
                                  force_collectives: bool = False, force_mode: str = None):
    """`formation_pass` for a swarm on the context's prepared time grid, from the waypoints: the solve and the sampler
    are ONE launch (`compute.solve_grid_sample`, msnap_solve_grid_sample_device), the rest is `formation_pass`.
    Returns (FormationResult, coef, dur, status).  `check_status` reads the solve's status back before the pairwise
    pass (one synchronisation) and refuses failed drones as `formation_pass` does; a caller that checks `status`
    itself later passes False and keeps the stream running."""
    lo, hi = shard_bounds(n_total, world, rank)
    multi = world > 1 or force_collectives
    coef, dur, status, pos, rows_t = compute.solve_grid_sample(wp_local, dt, n_samples, n_cols=None if multi else n_total)
    res = formation_pass(compute, coef, dur, n_total, world, rank, dt, n_samples, radius, dist=dist, torch=torch,
                         status_local=status if check_status else None, mesh_tris=mesh_tris,
                         force_collectives=force_collectives, force_mode=force_mode, _sampled=(pos, rows_t))
    return res, coef, dur, status


def default_sample_count(total_duration: float, dt: float) -> int:
    """len(np.arange(0, duration, dt)) -- the reference's sampling loop
    (src/trajectory_visualising/visualization.py:53)."""
    return int(len(np.arange(0.0, total_duration, dt)))
