"""Binary STL reader for the resources/stl obstacle meshes (host-side I/O).

80-byte header, little-endian u32 triangle count, 50 bytes per triangle
(normal 3 x f32, vertices 9 x f32, u16 attribute).  The reference loads these
with numpy-stl (src/RigidBodyPlanners/fcl_checker.py:19-20)."""
from __future__ import annotations

import numpy as np

_REC = np.dtype([("normal", "<f4", 3), ("v", "<f4", (3, 3)), ("attr", "<u2")])


def load_stl(path: str) -> np.ndarray:
    """-> float64 [n_tris, 3, 3] vertices."""
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 84:
        raise ValueError(f"{path}: too short for a binary STL")
    n = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
    if len(raw) < 84 + 50 * n:
        raise ValueError(f"{path}: header announces {n} triangles, file holds fewer")
    body = np.frombuffer(raw, dtype=_REC, count=n, offset=84)
    return np.ascontiguousarray(body["v"].astype(np.float64))


def load_stl_planner(path: str) -> np.ndarray:
    """As the planner's Fcl_mesh.load_stl sees the file: vertices rounded to 2 decimals
    (src/RigidBodyPlanners/fcl_checker.py:19-25, np.around(env_mesh.vectors, 2))."""
    return np.around(load_stl(path), 2)


def save_stl(path: str, tris: np.ndarray) -> None:
    """Write float vertices [n,3,3] as binary STL (normals recomputed, attribute 0)."""
    tris = np.asarray(tris, dtype=np.float64)
    rec = np.zeros(tris.shape[0], dtype=_REC)
    nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0])
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    rec["normal"] = np.where(ln > 0, nrm / np.where(ln > 0, ln, 1.0), 0.0)
    rec["v"] = tris
    with open(path, "wb") as f:
        f.write(b"msnap".ljust(80, b"\0"))
        f.write(np.uint32(tris.shape[0]).tobytes())
        f.write(rec.tobytes())


def box_mesh(lo, hi) -> np.ndarray:
    """12-triangle axis-aligned box, e.g. the wall of env-scene-ltu-experiment.stl
    ((-2,3.9,0) -> (2,4.1,1.6))."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    c = np.array([[lo[0] if (i >> 0) & 1 == 0 else hi[0],
                   lo[1] if (i >> 1) & 1 == 0 else hi[1],
                   lo[2] if (i >> 2) & 1 == 0 else hi[2]] for i in range(8)])
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tris = []
    for a, b, cc, d in quads:
        tris.append([c[a], c[b], c[cc]])
        tris.append([c[a], c[cc], c[d]])
    return np.array(tris)
