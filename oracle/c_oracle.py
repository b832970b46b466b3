"""ctypes loader of oracle/libmsnap_oracle.so (the C restatement).  TEST
INFRASTRUCTURE ONLY -- see msnap_oracle.c."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSNAP_ORACLE_LIB: the sanitizer build of the same file (oracle/Makefile `sanitize`, tests/test_sanitizers.py)
_PATH = os.environ.get("MSNAP_ORACLE_LIB") or os.path.join(_HERE, "libmsnap_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_PATH)
        lib.msnap_oracle_solve_batch.restype = ctypes.c_int
        lib.msnap_oracle_solve_batch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + \
            [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
        lib.msnap_oracle_max_threads.restype = ctypes.c_int
        lib.msnap_oracle_sample.restype = None
        lib.msnap_oracle_sample.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_double] + \
            [ctypes.c_int] * 3 + [ctypes.c_void_p]
        lib.msnap_oracle_formation_collide.restype = None
        lib.msnap_oracle_formation_collide.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_double,
                                                       ctypes.c_int] + [ctypes.c_void_p] * 3
        lib.msnap_oracle_mesh_sweep.restype = None
        lib.msnap_oracle_mesh_sweep.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 2
        _lib = lib
    return _lib


def max_threads() -> int:
    return int(load().msnap_oracle_max_threads())


def solve_batch(wp, t, ncoef=8, faithful=True, n_threads=1):
    """wp [N,m,4], t [N,m] or [m] -> coef [N,M,4,ncoef], dur [N,M], info [N], threads_used."""
    lib = load()
    wp = np.ascontiguousarray(wp, dtype=np.float64)
    t = np.ascontiguousarray(t, dtype=np.float64)
    N, m, _ = wp.shape
    M = m - 1
    coef = np.empty((N, M, 4, ncoef))
    dur = np.empty((N, M))
    info = np.empty((N,), dtype=np.int32)
    used = lib.msnap_oracle_solve_batch(N, M, ncoef, wp.ctypes.data, t.ctypes.data, int(t.ndim == 1),
                                        int(bool(faithful)), int(n_threads), coef.ctypes.data, dur.ctypes.data,
                                        info.ctypes.data)
    return coef, dur, info, int(used)


def sample_positions(coef, dur, dt, n_samples, naxes=3, n_threads=0):
    """coef [N,M,4,ncoef], dur [N,M] -> pos [N,S,naxes] (msnap_oracle.sample_positions in C)."""
    lib = load()
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    dur = np.ascontiguousarray(dur, dtype=np.float64)
    N, M = dur.shape
    pos = np.empty((N, int(n_samples), int(naxes)))
    lib.msnap_oracle_sample(N, M, coef.shape[3], coef.ctypes.data, dur.ctypes.data, float(dt), int(n_samples),
                            int(naxes), int(n_threads), pos.ctypes.data)
    return pos


def formation_collide(pos, radius, n_threads=0):
    """pos [N,S,3] -> (min_dist [N], partner [N] int32, hit [N] bool) (msnap_oracle.formation_collide in C)."""
    lib = load()
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    N, S, _ = pos.shape
    md = np.empty((N,))
    partner = np.empty((N,), dtype=np.int32)
    hit = np.empty((N,), dtype=np.int32)
    lib.msnap_oracle_formation_collide(N, S, pos.ctypes.data, float(radius), int(n_threads), md.ctypes.data,
                                       partner.ctypes.data, hit.ctypes.data)
    return md, partner, hit.astype(bool)


def mesh_sweep(pos, tris, radius, n_threads=0):
    """pos [N,S,3], tris [T,3,3] -> (min_dist [N], hit [N] bool) (msnap_oracle.mesh_sweep in C)."""
    lib = load()
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    tris = np.ascontiguousarray(tris, dtype=np.float64)
    N, S, _ = pos.shape
    md = np.empty((N,))
    hit = np.empty((N,), dtype=np.int32)
    lib.msnap_oracle_mesh_sweep(N, S, pos.ctypes.data, tris.shape[0], tris.ctypes.data, float(radius),
                                int(n_threads), md.ctypes.data, hit.ctypes.data)
    return md, hit.astype(bool)
