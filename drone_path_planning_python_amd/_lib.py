"""ctypes binding of libmsnap.so (include/msnap.h).

The shared library is the product: when it is missing or cannot be loaded this
module raises -- there is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSNAP_LIB_PATH: experiment builds only (e.g. a timing variant of the same ABI)
LIB_PATH = os.environ.get("MSNAP_LIB_PATH") or os.path.join(_HERE, "csrc", "libmsnap.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# symbol -> (restype, argtypes); mirrors include/msnap.h one to one
_VP = ctypes.c_void_p
_I = ctypes.c_int
_D = ctypes.c_double
SIGNATURES = {
    "msnap_version": (_I, []),
    "msnap_strerror": (ctypes.c_char_p, [_I]),
    "msnap_last_hip_error": (ctypes.c_char_p, [_VP]),
    "msnap_last_kernel": (ctypes.c_char_p, [_VP]),
    "msnap_create": (_I, [c_void_pp, _I, _I, _I]),
    "msnap_destroy": (None, [_VP]),
    "msnap_release_graph_buffers": (_I, [_VP, ctypes.POINTER(ctypes.c_size_t)]),
    "msnap_set_stream": (_I, [_VP, _VP]),
    "msnap_use_own_stream": (_I, [_VP]),
    "msnap_set_option": (_I, [_VP, ctypes.c_char_p, ctypes.c_long]),
    "msnap_get_option": (_I, [_VP, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)]),
    "msnap_host_alloc": (_I, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]),
    "msnap_host_free": (_I, [_VP]),
    "msnap_get_stream": (_VP, [_VP]),
    "msnap_sync": (_I, [_VP]),
    "msnap_timer_start": (_I, [_VP]),
    "msnap_timer_stop": (_I, [_VP, c_float_p]),
    "msnap_solve_batch": (_I, [_VP, _I, _I, _VP, _VP, _I, _VP, _VP, _VP]),
    "msnap_solve_batch_device": (_I, [_VP, _I, _I, _VP, _VP, _I, _VP, _VP, _VP]),
    "msnap_grid_prepare": (_I, [_VP, _I, _VP]),
    "msnap_grid_prepare_device": (_I, [_VP, _I, _VP]),
    "msnap_grid_segments": (_I, [_VP]),
    "msnap_solve_grid": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP]),
    "msnap_solve_grid_device": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP]),
    "msnap_pack_pol_matrix": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "msnap_pack_pol_matrix_device": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "msnap_formation_transform": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "msnap_formation_transform_device": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "msnap_sample": (_I, [_VP, _I, _I, _VP, _VP, _D, _I, _I, _VP]),
    "msnap_sample_device": (_I, [_VP, _I, _I, _VP, _VP, _D, _I, _I, _VP]),
    "msnap_eval_flat": (_I, [_VP, _I, _I, _VP, _VP, _I, _VP, _VP]),
    "msnap_eval_flat_device": (_I, [_VP, _I, _I, _VP, _VP, _I, _VP, _VP]),
    "msnap_snap_cost": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "msnap_snap_cost_device": (_I, [_VP, _I, _I, _VP, _VP, _VP]),
    "msnap_formation_collide": (_I, [_VP, _I, _I, _I, _I, _VP, _VP, _D, _VP, _VP, _VP]),
    "msnap_formation_collide_device": (_I, [_VP, _I, _I, _I, _I, _VP, _VP, _D, _VP, _VP, _VP]),
    "msnap_collide_rows_t_doubles": (ctypes.c_size_t, [_I, _I]),
    "msnap_formation_collide_reads_rows_t": (_I, [_VP, _I, _I, _I, _I]),
    "msnap_sample_collide_device": (_I, [_VP, _I, _I, _VP, _VP, _D, _I, _VP, _VP]),
    "msnap_solve_grid_sample_device": (_I, [_VP, _I, _I, _VP, _D, _I, _VP, _VP, _VP, _VP, _VP]),
    "msnap_formation_collide_t_device": (_I, [_VP, _I, _I, _I, _I, _VP, _VP, _VP, _D, _VP, _VP, _VP]),
    "msnap_formation_part_bytes": (ctypes.c_size_t, [_I]),
    "msnap_formation_collide_takes_broad_phase": (_I, [_VP, _I, _I, _I, _I]),
    "msnap_formation_whole_pass_pays": (_I, [_VP, _I, _I, ctypes.POINTER(_I)]),
    "msnap_formation_collide_part": (_I, [_VP, _I, _I, _VP, _I, _I, _VP]),
    "msnap_formation_collide_part_device": (_I, [_VP, _I, _I, _VP, _I, _I, _VP]),
    "msnap_formation_collide_finish": (_I, [_VP, _I, _I, _VP, _I, _I, _D, _VP, _VP, _VP]),
    "msnap_formation_collide_finish_device": (_I, [_VP, _I, _I, _VP, _I, _I, _D, _VP, _VP, _VP]),
    "msnap_mesh_sweep": (_I, [_VP, _I, _I, _VP, _I, _VP, _D, _VP, _VP]),
    "msnap_mesh_sweep_device": (_I, [_VP, _I, _I, _VP, _I, _VP, _D, _VP, _VP]),
    "msnap_mesh_validity": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _VP]),
    "msnap_mesh_validity_device": (_I, [_VP, _I, _VP, _I, _VP, _I, _VP, _VP]),
}


class MsnapError(RuntimeError):
    """A libmsnap entry point returned a negative msnap_error."""

    def __init__(self, code: int, message: str):
        super().__init__(f"msnap error {code}: {message}")
        self.code = code


_lib = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """Load libmsnap.so once.  Raises OSError with build instructions if absent."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C drone_path_planning_python_amd/csrc` (hipcc, gfx950). "
                "There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the ABI drifted
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(lib, ctx_handle, rc: int) -> None:
    if rc == 0:
        return
    msg = lib.msnap_strerror(rc).decode()
    if rc == -2 and ctx_handle:
        msg += ": " + lib.msnap_last_hip_error(ctx_handle).decode()
    raise MsnapError(rc, msg)


def csrc_sha() -> str:
    """sha256 (first 16 hex digits) over csrc/*.hip and csrc/*.h, in name order: the identity of the kernel
    sources.  Committed PMC summaries carry it (tools/pmc_summary.py, tools/make_pipeline_profile.sh) and
    bench.py only quotes counters whose value matches the sources it runs."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]
