"""Seeded synthetic swarms shared by bench.py and the tests."""
from __future__ import annotations

import numpy as np

BASE_SEED = 20260104


def swarm(config_id: int, n_drones: int, n_seg: int, shared_times: bool = False, rank: int = 0):
    """wp [N, M+1, 4] with x,y,z ~ U(-5,5) m and yaw ~ U(-pi,pi);
    times: per-drone T ~ U(0.5, 2.0) s (cumulative, t[0] = 0) or the reference's
    shared grid t_i = i*10/(M+1) (scripts/drones_pols_generator.py:44-46,56)."""
    rng = np.random.default_rng(BASE_SEED + config_id + 1000003 * rank)
    wp = np.empty((n_drones, n_seg + 1, 4))
    wp[..., :3] = rng.uniform(-5.0, 5.0, size=(n_drones, n_seg + 1, 3))
    wp[..., 3] = rng.uniform(-np.pi, np.pi, size=(n_drones, n_seg + 1))
    if shared_times:
        t = np.arange(n_seg + 1) * (10.0 / (n_seg + 1))
    else:
        T = rng.uniform(0.5, 2.0, size=(n_drones, n_seg))
        t = np.concatenate([np.zeros((n_drones, 1)), np.cumsum(T, axis=1)], axis=1)
    return wp, t


def formation_swarm(config_id: int, n_groups: int, n_seg: int, offsets: np.ndarray, rank: int = 0):
    """Formation-like swarm: n_groups random rigid-body paths, each carrying the
    K body-frame `offsets` (scripts/drones_traj_generator.py:22-38 has K = 2 at
    +-0.5 m).  Returns (rb_pose [G, M+1, 7], t [M+1] shared grid)."""
    rng = np.random.default_rng(BASE_SEED + config_id + 7919 * (rank + 1))
    m = n_seg + 1
    start = rng.uniform(-20.0, 20.0, size=(n_groups, 1, 3))
    step = rng.normal(0.0, 0.6, size=(n_groups, m, 3))
    pos = start + np.cumsum(step, axis=1)
    yaw = np.cumsum(rng.normal(0.0, 0.15, size=(n_groups, m)), axis=1)
    rb = np.zeros((n_groups, m, 7))
    rb[..., :3] = pos
    rb[..., 5] = np.sin(0.5 * yaw)   # rotation about z
    rb[..., 6] = np.cos(0.5 * yaw)
    t = np.arange(m) * (10.0 / m)
    return rb, t
