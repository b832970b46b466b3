"""Context: the Python face of one msnap_ctx (one GPU, one HIP stream).

Host entry points take / return NumPy arrays (the library stages them through
device buffers).  ``*_device`` entry points take objects exposing
``data_ptr()`` (torch CUDA tensors) and are asynchronous on the context's
stream; torch is plumbing for device memory only and is imported lazily.
"""
from __future__ import annotations

import ctypes
import threading

import numpy as np

from . import _lib

ST_OK, ST_SINGULAR, ST_TIMES, ST_NONFINITE = 0, 1, 2, 3
STATUS_TEXT = {
    ST_OK: "ok",
    ST_SINGULAR: "singular system",
    ST_TIMES: "times not strictly increasing",
    ST_NONFINITE: "non-finite waypoint or time",
}


def _ptr(x):
    """Device pointer of a torch tensor (or a raw int address)."""
    if x is None:
        return None
    if isinstance(x, int):
        return ctypes.c_void_p(x)
    return ctypes.c_void_p(x.data_ptr())


def _host(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(ctypes.c_void_p)


class _PinnedBlock:
    """Owner of one msnap_host_alloc block; arrays made by pinned_empty keep it alive."""

    def __init__(self, nbytes):
        self._lib = _lib.load()
        p = ctypes.c_void_p()
        _lib.check(self._lib, None, self._lib.msnap_host_alloc(ctypes.byref(p), max(int(nbytes), 1)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        ptr, self.ptr = getattr(self, "ptr", None), None
        if ptr:
            self._lib.msnap_host_free(ctypes.c_void_p(ptr))


def pinned_empty(shape, dtype=np.float64):
    """Uninitialised NumPy array in page-locked host memory (msnap_host_alloc).  Passing such
    arrays to solve_batch / solve_grid (inputs and `out=`) lets the copy engines move them
    directly, overlapped with the kernel; the memory is released with the last view."""
    dtype = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(int(s) for s in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    block = _PinnedBlock(nbytes)
    buf = (ctypes.c_char * max(nbytes, 1)).from_address(block.ptr)
    buf._msnap_owner = block          # the ctypes buffer is the array's base and holds the block
    return np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)


def _out_arrays(out, shapes):
    """Validate caller-provided (coef, dur, status) or allocate them."""
    dtypes = (np.float64, np.float64, np.int32)
    if out is None:
        return tuple(np.empty(s, dtype=d) for s, d in zip(shapes, dtypes))
    if len(out) != 3:
        raise ValueError("out must be (coef, dur, status)")
    for a, s, d in zip(out, shapes, dtypes):
        if not isinstance(a, np.ndarray) or a.shape != tuple(s) or a.dtype != d or not a.flags.c_contiguous \
                or not a.flags.writeable:
            raise ValueError(f"out arrays must be writable C-contiguous {np.dtype(d).name} of shape {tuple(s)}")
    return tuple(out)


class Context:
    """One msnap context (include/msnap.h: msnap_create / msnap_destroy)."""

    def __init__(self, device_id: int = 0, order: int = 7, max_segments: int = 256):
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        self._lock = threading.Lock()   # a context is not re-entrant (msnap.h)
        self._glock = threading.RLock()  # the prepared grid is context state: prepare + solve as one step
        rc = self._lib.msnap_create(ctypes.byref(self._h), int(device_id), int(order), int(max_segments))
        if rc != 0:
            self._h = ctypes.c_void_p()
            _lib.check(self._lib, None, rc)
        self.device_id = int(device_id)
        self.order = int(order)
        self.ncoef = self.order + 1
        self.max_segments = int(max_segments)

    # ---- lifetime -------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.msnap_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        _lib.check(self._lib, self._h, rc)

    # ---- launch-geometry options (msnap_set_option) ------------------------------
    def set_option(self, name: str, value: int):
        """Launch-geometry knob of this context (include/msnap.h lists the names), e.g.
        set_option("solve_grid_waves", 5) makes every wave of the large-batch solve walk
        several tiles on a batch small enough for the oracle."""
        with self._lock:
            self._ck(self._lib.msnap_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = ctypes.c_long()
        self._ck(self._lib.msnap_get_option(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def last_kernel(self) -> str:
        """Kernel instance the last solve call of this context launched (msnap_last_kernel)."""
        return self._lib.msnap_last_kernel(self._h).decode()

    def _coef_dur(self, coef, dur):
        """Validated host (coef [N,M,4,ncoef], dur [N,M]) and their pointers."""
        coef, pc = _host(coef, np.float64)
        dur, pd = _host(dur, np.float64)
        if dur.ndim != 2:
            raise ValueError("dur must be [N, M]")
        N, M = dur.shape
        if coef.shape != (N, M, 4, self.ncoef):
            raise ValueError(f"coef must be [N, M, 4, {self.ncoef}] = {(N, M, 4, self.ncoef)} for this "
                             f"order-{self.order} context, got {coef.shape}")
        return coef, pc, dur, pd, N, M

    # ---- stream / timing ------------------------------------------------------
    def set_stream(self, hip_stream: int | None):
        """Launch on an external hipStream_t (0 / None = the HIP null stream)."""
        with self._lock:
            self._ck(self._lib.msnap_set_stream(self._h, ctypes.c_void_p(hip_stream or 0)))

    def use_own_stream(self):
        with self._lock:
            self._ck(self._lib.msnap_use_own_stream(self._h))

    def stream(self) -> int:
        return int(self._lib.msnap_get_stream(self._h) or 0)

    def sync(self):
        with self._lock:
            self._ck(self._lib.msnap_sync(self._h))

    def timer_start(self):
        self._ck(self._lib.msnap_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = ctypes.c_float()
        self._ck(self._lib.msnap_timer_stop(self._h, ctypes.byref(ms)))
        return float(ms.value)

    # ---- a1/a2 solve ----------------------------------------------------------
    def solve_batch(self, wp, t, out=None):
        """wp [N, m, 4], t [N, m] or shared [m] (host) ->
        coef [N, M, 4, ncoef], dur [N, M], status [N] int32.
        `out=(coef, dur, status)` writes into caller-owned arrays (e.g. from pinned_empty)."""
        wp, pwp = _host(wp, np.float64)
        t, pt = _host(t, np.float64)
        if wp.ndim != 3 or wp.shape[2] != 4:
            raise ValueError("wp must be [N, m, 4]")
        N, m, _ = wp.shape
        shared = int(t.ndim == 1)
        if (shared and t.shape != (m,)) or (not shared and t.shape != (N, m)):
            raise ValueError("t must be [N, m] or [m]")
        M = m - 1
        coef, dur, status = _out_arrays(out, ((N, max(M, 0), 4, self.ncoef), (N, max(M, 0)), (N,)))
        with self._lock:
            self._ck(self._lib.msnap_solve_batch(
                self._h, N, M, pwp, pt, shared, coef.ctypes.data_as(ctypes.c_void_p),
                dur.ctypes.data_as(ctypes.c_void_p), status.ctypes.data_as(ctypes.c_void_p)))
        return coef, dur, status

    def solve_batch_device(self, n_drones, n_seg, wp, t, shared_times, coef, dur, status):
        with self._lock:
            self._ck(self._lib.msnap_solve_batch_device(
                self._h, int(n_drones), int(n_seg), _ptr(wp), _ptr(t), int(bool(shared_times)),
                _ptr(coef), _ptr(dur), _ptr(status)))

    # ---- shared time grid (K2: fp64 MFMA GEMM) ---------------------------------
    def prepare_grid(self, t):
        """Build the operator of the shared time grid t [m] (host array)."""
        t, pt = _host(t, np.float64)
        if t.ndim != 1 or t.shape[0] < 2:
            raise ValueError("t must be [m], m >= 2")
        with self._glock:
            with self._lock:
                self._ck(self._lib.msnap_grid_prepare(self._h, t.shape[0] - 1, pt))
            self._grid_host = t.copy()

    def ensure_grid(self, t):
        """prepare_grid(t) unless the context already holds exactly this grid."""
        t = np.ascontiguousarray(t, dtype=np.float64)
        with self._glock:
            g = getattr(self, "_grid_host", None)
            if g is None or g.shape != t.shape or not np.array_equal(g, t):
                self.prepare_grid(t)

    def solve_on_grid(self, t, wp, out=None):
        """ensure_grid(t) + solve_grid(wp) as one step: two threads sharing a context (the node's
        callback1 / callback2) cannot swap the grid between the two calls."""
        with self._glock:
            self.ensure_grid(t)
            return self.solve_grid(wp, out=out)

    def prepare_grid_device(self, n_seg, t):
        with self._glock:     # (under the grid lock like prepare_grid: solve_on_grid must not see a half-updated grid)
            with self._lock:
                self._ck(self._lib.msnap_grid_prepare_device(self._h, int(n_seg), _ptr(t)))
            self._grid_host = None

    def grid_segments(self) -> int:
        """Segments of the grid this context holds (msnap_grid_segments; 0: none prepared)."""
        n = int(self._lib.msnap_grid_segments(self._h))
        if n < 0:
            _lib.check(self._lib, self._h, n)
        return n

    def grid_waypoints(self) -> int:
        """Waypoints per drone of the grid this context was last prepared for (raises MSNAP_ENOGRID without one)."""
        n = self.grid_segments()
        if n == 0:
            _lib.check(self._lib, self._h, -7)
        return n + 1

    def release_graph_buffers(self) -> int:
        """Free the blocks kept alive for graphs captured before a buffer grew (msnap_release_graph_buffers): call
        once every graph that captured calls of this context is destroyed.  Returns the bytes released."""
        n = ctypes.c_size_t()
        with self._lock:
            self._ck(self._lib.msnap_release_graph_buffers(self._h, ctypes.byref(n)))
        return int(n.value)

    def solve_grid(self, wp, out=None):
        """wp [N, m, 4] on the prepared grid -> coef, dur, status (as solve_batch)."""
        with self._glock:
            wp, pwp = _host(wp, np.float64)
            m = self.grid_waypoints()
            if wp.ndim != 3 or wp.shape[1:] != (m, 4):
                raise ValueError(f"wp must be [N, {m}, 4] for the prepared grid")
            N, M = wp.shape[0], m - 1
            coef, dur, status = _out_arrays(out, ((N, M, 4, self.ncoef), (N, M), (N,)))
            with self._lock:
                self._ck(self._lib.msnap_solve_grid(self._h, N, M, pwp, coef.ctypes.data_as(ctypes.c_void_p),
                                                   dur.ctypes.data_as(ctypes.c_void_p),
                                                   status.ctypes.data_as(ctypes.c_void_p)))
            return coef, dur, status

    def solve_grid_device(self, n_drones, n_seg, wp, coef, dur, status):
        """Device pointers.  `n_seg` is the segment count the buffers were sized for: the library refuses
        (MSNAP_ESEGMENTS) a count that is not the prepared grid's instead of writing past them."""
        with self._lock:
            self._ck(self._lib.msnap_solve_grid_device(self._h, int(n_drones), int(n_seg), _ptr(wp), _ptr(coef),
                                                      _ptr(dur), _ptr(status)))

    # ---- a7 pack ---------------------------------------------------------------
    def pack_pol_matrix(self, coef, dur):
        coef, pc, dur, pd, N, M = self._coef_dur(coef, dur)
        out = np.empty((N, M, 1 + 4 * self.ncoef), dtype=np.float32)
        with self._lock:
            self._ck(self._lib.msnap_pack_pol_matrix(self._h, N, M, pc, pd, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def pack_pol_matrix_device(self, n_drones, n_seg, coef, dur, out):
        with self._lock:
            self._ck(self._lib.msnap_pack_pol_matrix_device(self._h, int(n_drones), int(n_seg), _ptr(coef),
                                                           _ptr(dur), _ptr(out)))

    # ---- a8 formation transform -----------------------------------------------
    def formation_transform(self, rb_pose, offsets):
        rb, prb = _host(rb_pose, np.float64)
        off, poff = _host(offsets, np.float64)
        if rb.ndim != 2 or rb.shape[1] != 7 or off.ndim != 2 or off.shape[1] != 3:
            raise ValueError("rb_pose must be [P, 7], offsets [K, 3]")
        P, K = rb.shape[0], off.shape[0]
        out = np.empty((K, P, 7), dtype=np.float64)
        with self._lock:
            self._ck(self._lib.msnap_formation_transform(self._h, P, K, prb, poff,
                                                        out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def formation_transform_device(self, n_poses, n_offsets, rb_pose, offsets, out):
        with self._lock:
            self._ck(self._lib.msnap_formation_transform_device(self._h, int(n_poses), int(n_offsets),
                                                               _ptr(rb_pose), _ptr(offsets), _ptr(out)))

    # ---- a5 sampler ------------------------------------------------------------
    def sample(self, coef, dur, dt: float, n_samples: int, n_axes: int = 3):
        coef, pc, dur, pd, N, M = self._coef_dur(coef, dur)
        pos = np.empty((N, int(n_samples), int(n_axes)), dtype=np.float64)
        with self._lock:
            self._ck(self._lib.msnap_sample(self._h, N, M, pc, pd, float(dt), int(n_samples), int(n_axes),
                                           pos.ctypes.data_as(ctypes.c_void_p)))
        return pos

    def sample_device(self, n_drones, n_seg, coef, dur, dt, n_samples, n_axes, pos):
        with self._lock:
            self._ck(self._lib.msnap_sample_device(self._h, int(n_drones), int(n_seg), _ptr(coef), _ptr(dur),
                                                  float(dt), int(n_samples), int(n_axes), _ptr(pos)))

    # ---- flatness evaluator (Trajectory.eval) -------------------------------------
    def eval_flat(self, coef, dur, ts):
        """coef [N,M,4,nc], dur [N,M], ts [S] -> out [N,S,13] = pos3 vel3 acc3 omega3 yaw."""
        coef, pc, dur, pd, N, M = self._coef_dur(coef, dur)
        ts, pt = _host(ts, np.float64)
        if ts.ndim != 1:
            raise ValueError("ts must be [S]")
        S = ts.shape[0]
        out = np.empty((N, S, 13), dtype=np.float64)
        with self._lock:
            self._ck(self._lib.msnap_eval_flat(self._h, N, M, pc, pd, S, pt, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def eval_flat_device(self, n_drones, n_seg, coef, dur, n_samples, ts, out):
        with self._lock:
            self._ck(self._lib.msnap_eval_flat_device(self._h, int(n_drones), int(n_seg), _ptr(coef), _ptr(dur),
                                                     int(n_samples), _ptr(ts), _ptr(out)))

    # ---- snap cost ---------------------------------------------------------------------
    def snap_cost(self, coef, dur):
        """J = sum_seg int (p^(k))^2 dt per drone and axis -> [N, 4]."""
        coef, pc, dur, pd, N, M = self._coef_dur(coef, dur)
        cost = np.empty((N, 4), dtype=np.float64)
        with self._lock:
            self._ck(self._lib.msnap_snap_cost(self._h, N, M, pc, pd, cost.ctypes.data_as(ctypes.c_void_p)))
        return cost

    def snap_cost_device(self, n_drones, n_seg, coef, dur, cost):
        with self._lock:
            self._ck(self._lib.msnap_snap_cost_device(self._h, int(n_drones), int(n_seg), _ptr(coef), _ptr(dur),
                                                     _ptr(cost)))

    # ---- collision passes --------------------------------------------------------
    def formation_collide(self, pos_rows, pos_cols, radius: float, row_offset: int = 0):
        pr, ppr = _host(pos_rows, np.float64)
        pc, ppc = _host(pos_cols, np.float64)
        if pr.ndim != 3 or pr.shape[2] != 3 or pc.ndim != 3 or pc.shape[1:] != pr.shape[1:]:
            raise ValueError("pos_rows [R,S,3] and pos_cols [C,S,3] must share S")
        R, S, _ = pr.shape
        Cn = pc.shape[0]
        md = np.empty((R,), dtype=np.float64)
        partner = np.empty((R,), dtype=np.int32)
        hit = np.empty((R,), dtype=np.int32)
        with self._lock:
            self._ck(self._lib.msnap_formation_collide(
                self._h, R, int(row_offset), Cn, S, ppr, ppc, float(radius),
                md.ctypes.data_as(ctypes.c_void_p), partner.ctypes.data_as(ctypes.c_void_p),
                hit.ctypes.data_as(ctypes.c_void_p)))
        return md, partner, hit.astype(bool)

    def formation_collide_device(self, n_rows, row_offset, n_cols, n_samples, pos_rows, pos_cols, radius,
                                 min_dist, partner, hit):
        with self._lock:
            self._ck(self._lib.msnap_formation_collide_device(
                self._h, int(n_rows), int(row_offset), int(n_cols), int(n_samples), _ptr(pos_rows),
                _ptr(pos_cols), float(radius), _ptr(min_dist), _ptr(partner), _ptr(hit)))

    def collide_rows_t_doubles(self, n_rows: int, n_samples: int) -> int:
        return int(self._lib.msnap_collide_rows_t_doubles(int(n_rows), int(n_samples)))

    def collide_takes_broad_phase(self, n_rows: int, row_offset: int, n_cols: int, n_samples: int) -> bool:
        """Whether formation_collide with these arguments would run behind the exact broad phase (the library's own
        predicate: size limits and the "collide_*" options)."""
        return bool(self._lib.msnap_formation_collide_takes_broad_phase(self._h, int(n_rows), int(row_offset),
                                                                        int(n_cols), int(n_samples)))

    def whole_pass_pays(self, n_drones: int, n_ranks: int) -> bool:
        """After a whole-swarm pass: does running it on every one of `n_ranks` ranks beat dividing the pairs (the
        library's cost model on the counts that pass left; synchronises the stream)?"""
        pays = ctypes.c_int()
        with self._lock:
            self._ck(self._lib.msnap_formation_whole_pass_pays(self._h, int(n_drones), int(n_ranks), ctypes.byref(pays)))
        return bool(pays.value)

    def collide_reads_rows_t(self, n_rows: int, row_offset: int, n_cols: int, n_samples: int) -> bool:
        """Whether formation_collide_t_device with these arguments would read a row image (msnap.h): a whole swarm
        behind the exact broad phase builds its own, spatially sorted one."""
        return bool(self._lib.msnap_formation_collide_reads_rows_t(self._h, int(n_rows), int(row_offset), int(n_cols),
                                                                   int(n_samples)))

    def sample_collide_device(self, n_drones, n_seg, coef, dur, dt, n_samples, pos, pos_t):
        """The sampler with its second output: the transposed row image the pairwise pass reads."""
        with self._lock:
            self._ck(self._lib.msnap_sample_collide_device(self._h, int(n_drones), int(n_seg), _ptr(coef), _ptr(dur),
                                                          float(dt), int(n_samples), _ptr(pos), _ptr(pos_t)))

    def solve_grid_sample_device(self, n_drones, n_seg, wp, dt, n_samples, coef, dur, status, pos, pos_t=None):
        """The shared-grid solve and the sampler as one launch (msnap.h): the outputs of solve_grid_device followed by
        sample_collide_device (`pos_t` given) or the 3-axis sampler, bit for bit."""
        with self._lock:
            self._ck(self._lib.msnap_solve_grid_sample_device(
                self._h, int(n_drones), int(n_seg), _ptr(wp), float(dt), int(n_samples), _ptr(coef), _ptr(dur),
                _ptr(status), _ptr(pos), _ptr(pos_t) if pos_t is not None else None))

    def formation_collide_t_device(self, n_rows, row_offset, n_cols, n_samples, pos_rows_t, pos_rows, pos_cols, radius,
                                   min_dist, partner, hit):
        with self._lock:
            self._ck(self._lib.msnap_formation_collide_t_device(
                self._h, int(n_rows), int(row_offset), int(n_cols), int(n_samples), _ptr(pos_rows_t), _ptr(pos_rows),
                _ptr(pos_cols), float(radius), _ptr(min_dist), _ptr(partner), _ptr(hit)))

    # ---- the pairwise pass in parts: every unordered pair on exactly one rank (include/msnap.h) ----
    def formation_part_bytes(self, n_drones: int) -> int:
        return int(self._lib.msnap_formation_part_bytes(int(n_drones)))

    def formation_collide_part(self, pos_all, part: int, n_parts: int):
        """One rank's part of the pass over the whole swarm `pos_all` [N,S,3]: a uint8 block of
        formation_part_bytes(N) bytes (squared minima [N] float64, then partners [N] int32)."""
        pa, ppa = _host(pos_all, np.float64)
        if pa.ndim != 3 or pa.shape[2] != 3:
            raise ValueError("pos_all must be [N, S, 3]")
        N, S, _ = pa.shape
        out = np.empty((self.formation_part_bytes(N),), dtype=np.uint8)
        with self._lock:
            self._ck(self._lib.msnap_formation_collide_part(self._h, N, S, ppa, int(part), int(n_parts),
                                                            out.ctypes.data_as(ctypes.c_void_p)))
        return out

    @staticmethod
    def split_part(block, n_drones: int):
        """(squared minima [N] float64, partners [N] int32) views of one part block."""
        b = np.ascontiguousarray(block, dtype=np.uint8)
        return b[:8 * n_drones].view(np.float64), b[8 * n_drones:12 * n_drones].view(np.int32)

    def formation_collide_finish(self, parts, n_drones: int, radius: float, row_offset: int = 0, n_rows=None):
        """Fold the parts (uint8 [P, formation_part_bytes(N)]) for the rows [row_offset, row_offset + n_rows)."""
        pb, ppb = _host(parts, np.uint8)
        stride = self.formation_part_bytes(n_drones)
        if pb.ndim != 2 or pb.shape[1] != stride:
            raise ValueError("parts must be [P, formation_part_bytes(N)] uint8")
        n_rows = n_drones - row_offset if n_rows is None else int(n_rows)
        md = np.empty((n_rows,), dtype=np.float64)
        partner = np.empty((n_rows,), dtype=np.int32)
        hit = np.empty((n_rows,), dtype=np.int32)
        with self._lock:
            self._ck(self._lib.msnap_formation_collide_finish(
                self._h, int(n_drones), pb.shape[0], ppb, int(row_offset), n_rows, float(radius),
                md.ctypes.data_as(ctypes.c_void_p), partner.ctypes.data_as(ctypes.c_void_p),
                hit.ctypes.data_as(ctypes.c_void_p)))
        return md, partner, hit.astype(bool)

    def formation_collide_part_device(self, n_drones, n_samples, pos_all, part, n_parts, part_out):
        with self._lock:
            self._ck(self._lib.msnap_formation_collide_part_device(
                self._h, int(n_drones), int(n_samples), _ptr(pos_all), int(part), int(n_parts), _ptr(part_out)))

    def formation_collide_finish_device(self, n_drones, n_parts, parts, row_offset, n_rows, radius, min_dist,
                                        partner, hit):
        with self._lock:
            self._ck(self._lib.msnap_formation_collide_finish_device(
                self._h, int(n_drones), int(n_parts), _ptr(parts), int(row_offset), int(n_rows), float(radius),
                _ptr(min_dist), _ptr(partner), _ptr(hit)))

    def mesh_sweep(self, pos, tris, radius: float):
        p, pp = _host(pos, np.float64)
        tr, ptr_ = _host(tris, np.float64)
        if p.ndim != 3 or p.shape[2] != 3:
            raise ValueError("pos must be [N, S, 3]")
        if tr.ndim != 3 or tr.shape[1:] != (3, 3):
            raise ValueError("tris must be [T, 3, 3]")
        N, S, _ = p.shape
        Tn = tr.shape[0]
        md = np.empty((N,), dtype=np.float64)
        hit = np.empty((N,), dtype=np.int32)
        with self._lock:
            self._ck(self._lib.msnap_mesh_sweep(self._h, N, S, pp, Tn, ptr_, float(radius),
                                               md.ctypes.data_as(ctypes.c_void_p),
                                               hit.ctypes.data_as(ctypes.c_void_p)))
        return md, hit.astype(bool)

    def mesh_validity(self, states, robot_tris, env_tris):
        """states [N,4] (x,y,z,yaw), meshes [R,3,3] / [E,3,3] -> valid [N] bool (True = no collision):
        the planner's isStateValid for a batch of states."""
        st, pst = _host(states, np.float64)
        rt, prt = _host(robot_tris, np.float64)
        et, pet = _host(env_tris, np.float64)
        if st.ndim != 2 or st.shape[1] != 4:
            raise ValueError("states must be [N, 4]")
        if rt.ndim != 3 or rt.shape[1:] != (3, 3) or et.ndim != 3 or et.shape[1:] != (3, 3):
            raise ValueError("robot_tris / env_tris must be [T, 3, 3]")
        N = st.shape[0]
        valid = np.empty((N,), dtype=np.int32)
        with self._lock:
            self._ck(self._lib.msnap_mesh_validity(self._h, N, pst, rt.shape[0], prt, et.shape[0], pet,
                                                  valid.ctypes.data_as(ctypes.c_void_p)))
        return valid.astype(bool)

    def mesh_validity_device(self, n_states, states, n_rtris, rtris, n_etris, etris, valid):
        with self._lock:
            self._ck(self._lib.msnap_mesh_validity_device(self._h, int(n_states), _ptr(states), int(n_rtris),
                                                         _ptr(rtris), int(n_etris), _ptr(etris), _ptr(valid)))

    def mesh_sweep_device(self, n_drones, n_samples, pos, n_tris, tris, radius, min_dist, hit):
        with self._lock:
            self._ck(self._lib.msnap_mesh_sweep_device(self._h, int(n_drones), int(n_samples), _ptr(pos),
                                                      int(n_tris), _ptr(tris), float(radius), _ptr(min_dist),
                                                      _ptr(hit)))


_default = {}
_default_lock = threading.Lock()


def default_context(order: int = 7, device_id: int = 0) -> Context:
    """Process-wide context per (device, order), created on first use."""
    key = (int(device_id), int(order))
    with _default_lock:
        ctx = _default.get(key)
        if ctx is None:
            ctx = Context(device_id=device_id, order=order, max_segments=4096)
            _default[key] = ctx
        return ctx
