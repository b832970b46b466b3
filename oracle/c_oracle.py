"""ctypes loader of oracle/libmsnap_oracle.so (the C restatement).  TEST
INFRASTRUCTURE ONLY -- see msnap_oracle.c."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libmsnap_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(_PATH)
        lib.msnap_oracle_solve_batch.restype = ctypes.c_int
        lib.msnap_oracle_solve_batch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + \
            [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
        lib.msnap_oracle_max_threads.restype = ctypes.c_int
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
