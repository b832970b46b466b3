"""`drones_pols_generator` node: per-drone Path -> piecewise polynomial.

Keeps the reference node's API (scripts/drones_pols_generator.py): `callback1`,
`callback2` (each fires once, :22-37), `path_to_pol(path, cfid)` (:40-90),
`listener()` (:93-106), topic names 'drone1Path', 'drone2Path', 'piece_pol'.
rospy / tf are imported lazily, so `path_to_pol` also works on the ROS-free
stand-in messages of nodes/msgs.py.  The solve and the float32 pack run on the
GPU through libmsnap (Context.solve_batch / pack_pol_matrix); `paths_to_pols`
is the batched form for whole swarms.
"""
from __future__ import annotations

import math
import os
from typing import Sequence

import numpy as np

from ..context import Context, default_context
from ..optimizations.calculatingTrajectories import raise_for_status
from . import msgs

TOTAL_DURATION = 10.0   # secs, scripts/drones_pols_generator.py:44
OUTPUT_DIR = os.environ.get("MSNAP_TRAJ_DIR", os.path.join(os.getcwd(), "trajectories"))

piece_pols_pub = None   # set by listener(); tests may install any object with .publish()


def yaw_from_quaternion(q) -> float:
    """tf.transformations.euler_from_quaternion(q)[2] for the default 'sxyz' axes
    (reference :51-53): rotation matrix entries r00, r10 of the normalised
    quaternion, yaw = atan2(r10, r00) unless the pitch is at its singularity."""
    x, y, z, w = (float(v) for v in q)
    n = x * x + y * y + z * z + w * w
    eps = np.finfo(float).eps * 4.0
    if n < eps:
        return 0.0
    s = 2.0 / n
    r00 = 1.0 - s * (y * y + z * z)
    r10 = s * (x * y + z * w)
    if math.sqrt(r00 * r00 + r10 * r10) > eps:
        return math.atan2(r10, r00)
    return 0.0


def path_time_grid(n_poses: int, total_duration: float = TOTAL_DURATION) -> np.ndarray:
    """t_i = i * total/n (reference :44-46, :56; the last waypoint sits at total*(n-1)/n)."""
    step = total_duration / n_poses
    return np.array([step * i for i in range(n_poses)], dtype=np.float64)


def allocate_times(positions: np.ndarray, total_duration: float = TOTAL_DURATION, mode: str = "uniform") -> np.ndarray:
    """Waypoint times for a path of m poses.

    "uniform"  the reference's grid, t_i = i * total / m (reference :44-46,56);
    "distance" SURVEY.md 8f rank 3: segment durations proportional to the Euclidean
               length of each leg (zero-length legs get the mean share so that the times
               stay strictly increasing), scaled so that the last waypoint keeps the
               reference's end time total * (m-1) / m."""
    positions = np.asarray(positions, dtype=np.float64)
    m = positions.shape[0]
    if mode == "uniform":
        return path_time_grid(m, total_duration)
    if mode != "distance":
        raise ValueError("mode must be 'uniform' or 'distance'")
    leg = np.linalg.norm(np.diff(positions[:, :3], axis=0), axis=1)
    mean = leg.mean() if leg.size and leg.mean() > 0 else 1.0
    leg = np.where(leg > 0, leg, mean)
    end = total_duration * (m - 1) / m
    t = np.concatenate([[0.0], np.cumsum(leg)])
    return t * (end / t[-1])


def paths_to_waypoints(paths: Sequence) -> tuple:
    """Path-likes with equal pose counts -> (wp [N,m,4], t [m])."""
    arrays = [msgs.path_to_arrays(p) for p in paths]
    m = arrays[0][0].shape[0]
    wp = np.empty((len(paths), m, 4))
    for k, (pos, quat) in enumerate(arrays):
        if pos.shape[0] != m:
            raise ValueError("all paths of one batch must have the same number of poses")
        wp[k, :, :3] = pos
        wp[k, :, 3] = [yaw_from_quaternion(q) for q in quat]
    return wp, path_time_grid(m)


def paths_to_pols(paths: Sequence, ctx: Context | None = None):
    """Batched path_to_pol: returns (matrix f32 [N, M, 33], coef, dur)."""
    ctx = ctx or default_context(7)
    wp, t = paths_to_waypoints(paths)
    # every path of the node shares the uniform grid: one operator, one MFMA GEMM per batch
    coef, dur, status = ctx.solve_on_grid(t, wp)
    for k in range(len(paths)):
        raise_for_status(int(status[k]), t)
    return ctx.pack_pol_matrix(coef, dur), coef, dur


def build_message(matrix: np.ndarray, cfid: int):
    """Fill TrajectoryPolynomialPieceMarios from the [M,33] float32 matrix (reference :83-87)."""
    try:   # the real message package when a ROS workspace provides it (reference :11-14)
        from execution.msg import TrajectoryPolynomialPieceMarios as Msg   # type: ignore
    except Exception:
        try:
            from crazyswarm.msg import TrajectoryPolynomialPieceMarios as Msg   # type: ignore
        except Exception:
            Msg = msgs.TrajectoryPolynomialPieceMarios
    m = Msg()
    m.cf_id = cfid
    m.poly_x = list(matrix[:, 1:9].flatten())
    m.poly_y = list(matrix[:, 9:17].flatten())
    m.poly_z = list(matrix[:, 17:25].flatten())
    m.poly_yaw = list(matrix[:, 25:33].flatten())
    m.durations = list(matrix[:, 0].flatten())
    return m


def save_pol_matrix(matrix: np.ndarray, cfid: int, out_dir: str | None = None) -> str:
    """Pol_matrix_{cfid}.csv exactly as the reference writes it: bare np.savetxt,
    comma separated, '%.18e', no header (reference :79-81)."""
    out_dir = out_dir or OUTPUT_DIR
    os.makedirs(out_dir, exist_ok=True)
    fn = os.path.join(out_dir, "Pol_matrix_{}.csv".format(cfid))
    np.savetxt(fn, matrix, delimiter=",")
    return fn


def path_to_pol(path, cfid: int, ctx: Context | None = None, out_dir: str | None = None, save: bool = True):
    """One drone's Path -> polynomial pieces: solve, pack to float32, write the
    CSV, publish on 'piece_pol' (reference :40-90).  Returns the message."""
    print("Path received...")
    matrix, _, _ = paths_to_pols([path], ctx)
    matrix = matrix[0]
    if save:
        save_pol_matrix(matrix, cfid, out_dir)
    pol_to_send = build_message(matrix, cfid)
    if piece_pols_pub is not None:
        piece_pols_pub.publish(pol_to_send)
        print("Published polynomial piece...")
    return pol_to_send


def callback1(path):
    if callback1.counter == 0:
        path_to_pol(path, 1)
        callback1.counter += 1


callback1.counter = 0


def callback2(path):
    if callback2.counter == 0:
        path_to_pol(path, 2)
        callback2.counter += 1


callback2.counter = 0


def listener():
    """ROS entry point (reference :93-106); needs rospy + nav_msgs at run time."""
    global piece_pols_pub
    import rospy   # noqa: WPS433 (lazy: the library part must import without ROS)
    from nav_msgs.msg import Path
    rospy.init_node('drones_path_listener')
    sample = build_message(np.zeros((1, 33), dtype=np.float32), 0)
    piece_pols_pub = rospy.Publisher('piece_pol', type(sample), queue_size=10)
    rospy.Subscriber('drone1Path', Path, callback1)
    rospy.Subscriber('drone2Path', Path, callback2)
    rospy.spin()


if __name__ == '__main__':
    listener()
