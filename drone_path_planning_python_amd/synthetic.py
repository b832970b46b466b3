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


# ---------------------------------------------------------------------------------------
# BASELINE.json configs[2] / configs[3] at full size: formation-like swarms
# ---------------------------------------------------------------------------------------
# 8 drones per rigid body on a 1.0 x 1.0 x 0.5 m lattice: with spheres of radius 0.15 m team mates
# never touch (0.5 m > 2 r), so every hit is between different rigid bodies and hits stay sparse.
FORMATION_OFFSETS = np.array([[sx * 0.5, sy * 0.5, sz * 0.25] for sx in (1, -1) for sy in (1, -1) for sz in (1, -1)],
                             dtype=np.float64)
DRONE_RADIUS = 0.15     # y half-extent of resources/stl/custom_triangle_robot.stl
SAMPLE_DT = 0.1         # the reference's sampling step (scripts/path_vis.py:28)
FORMATION_SEGMENTS = {2: 10, 3: 20}


def formation_config(config_index: int, n_groups: int = 512):
    """Inputs of BASELINE.json configs[config_index] (2: 4096 drones x 10 segments + pairwise pass,
    3: 4096 drones x 20 segments + STL sweep): `n_groups` rigid-body paths of m = M + 1 poses on the
    reference's uniform grid t_i = i * 10 / m, each carrying FORMATION_OFFSETS through the a8
    transform (scripts/drones_traj_generator.py:56-89).  Returns (rb_pose [G, m, 7], offsets [K, 3],
    t [m]); drone g * K + k is offset k of rigid body g.

    configs[2]: bodies drift in random directions inside a 130 x 130 x 12 m box.
    configs[3]: bodies start south of the wall of resources/stl/env-scene-hole.stl (the plane
    y = 0, |x| < 4.09, |z| < 2.19, with its hole) and fly north past it and past the wall of
    env-scene-ltu-experiment.stl (y = 4, |x| < 2, 0 < z < 1.6): some cross a wall, most do not."""
    if config_index not in FORMATION_SEGMENTS:
        raise ValueError("config_index must be 2 or 3")
    M = FORMATION_SEGMENTS[config_index]
    m = M + 1
    rng = np.random.default_rng(BASE_SEED + 100 * config_index)
    G = n_groups
    s = np.linspace(0.0, 1.0, m)[None, :, None]
    if config_index == 2:
        start = rng.uniform([-65.0, -65.0, 0.0], [65.0, 65.0, 12.0], size=(G, 3))[:, None, :]
        heading = rng.uniform(-np.pi, np.pi, size=G)
        travel = rng.uniform(4.0, 9.0, size=G)
        vel = np.stack([np.cos(heading) * travel, np.sin(heading) * travel, rng.uniform(-1.0, 1.0, size=G)], axis=1)
    else:
        start = rng.uniform([-90.0, -22.0, -6.0], [90.0, -2.0, 12.0], size=(G, 3))[:, None, :]
        vel = np.stack([rng.uniform(-2.0, 2.0, size=G), rng.uniform(7.0, 11.0, size=G),
                        rng.uniform(-0.8, 0.8, size=G)], axis=1)
    wobble = np.cumsum(rng.normal(0.0, 0.12, size=(G, m, 3)), axis=1)
    pos = start + s * vel[:, None, :] + wobble
    yaw = np.arctan2(vel[:, 1], vel[:, 0])[:, None] + np.cumsum(rng.normal(0.0, 0.08, size=(G, m)), axis=1)
    rb = np.zeros((G, m, 7))
    rb[..., :3] = pos
    rb[..., 5] = np.sin(0.5 * yaw)   # rotation about z: quaternion_from_euler(0, 0, yaw)
    rb[..., 6] = np.cos(0.5 * yaw)
    t = np.arange(m) * (10.0 / m)
    return rb, FORMATION_OFFSETS.copy(), t


def formation_waypoints(poses: np.ndarray, n_groups: int) -> np.ndarray:
    """a8 output [K, G*m, 7] (the G paths' poses laid end to end) -> wp [G*K, m, 4] with drone
    g*K + k = offset k of body g; yaw = euler_from_quaternion(q)[2] as path_to_pol reads it
    (scripts/drones_pols_generator.py:51-53), vectorised."""
    K, P, _ = poses.shape
    m = P // n_groups
    p = poses.reshape(K, n_groups, m, 7).transpose(1, 0, 2, 3).reshape(n_groups * K, m, 7)
    x, y, z, w = p[..., 3], p[..., 4], p[..., 5], p[..., 6]
    n = x * x + y * y + z * z + w * w
    sc = 2.0 / n
    wp = np.empty((n_groups * K, m, 4))
    wp[..., :3] = p[..., :3]
    wp[..., 3] = np.arctan2(sc * (x * y + z * w), 1.0 - sc * (y * y + z * z))
    return wp


def formation_sample_count(t: np.ndarray, dt: float = SAMPLE_DT) -> int:
    """len(np.arange(0, duration, dt)) for the path's total duration t[-1] - t[0]
    (src/trajectory_visualising/visualization.py:53)."""
    return int(len(np.arange(0.0, float(t[-1] - t[0]), dt)))
