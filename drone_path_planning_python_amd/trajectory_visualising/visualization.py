"""`get_nav_path_msg`: a loaded trajectory sampled into a nav_msgs/Path.

Keeps the reference function's API and output (src/trajectory_visualising/visualization.py:39-71):
one pose per t in np.arange(0, tr.duration, timestep), position = Trajectory.eval(t).pos + offset,
orientation = tf.transformations.quaternion_from_euler(0, 0, -yaw), frame "world".  The evaluation of
all sample instants is one call of msnap_eval_flat (the flatness evaluator kernel); nothing is
evaluated on the host.  rospy / nav_msgs are used when a ROS installation provides them, otherwise
the stand-in messages of nodes/msgs.py (same attribute names).
"""
from __future__ import annotations

import math
import time
from typing import Sequence

import numpy as np

from ..context import Context, default_context
from ..nodes import msgs


def quaternion_from_yaw(yaw: float):
    """tf.transformations.quaternion_from_euler(0, 0, yaw) for the default 'sxyz' axes
    (reference :61-62 calls it with -out.yaw): with roll = pitch = 0 the half-angle products of the
    published algorithm reduce to (x, y, z, w) = (0, 0, sin(yaw / 2), cos(yaw / 2)).  tf is not vendored
    in the reference: parity unpinned."""
    h = 0.5 * float(yaw)
    return (0.0, 0.0, math.sin(h), math.cos(h))


def trajectory_arrays(tr) -> tuple:
    """Trajectory (list of Polynomial4D pieces) -> coef [1, M, 4, 8], dur [1, M]."""
    M = tr.n_pieces()
    coef = np.empty((1, M, 4, 8))
    dur = np.empty((1, M))
    for i, piece in enumerate(tr.polynomials):
        dur[0, i] = float(piece.duration)
        for a, pol in enumerate((piece.px, piece.py, piece.pz, piece.pyaw)):
            coef[0, i, a, :] = np.asarray(pol.p, dtype=np.float64).reshape(-1)
    return coef, dur


def _new_path(frame_id: str):
    try:   # the real message classes when a ROS workspace provides them (reference :1-5)
        import rospy
        from nav_msgs.msg import Path
        msg = Path()
        msg.header.frame_id = frame_id
        msg.header.stamp = rospy.Time.now()
        return msg
    except Exception:
        return msgs.Path(header=msgs.Header(frame_id=frame_id, stamp=time.time()))


def _new_pose():
    try:
        from geometry_msgs.msg import PoseStamped
        return PoseStamped()
    except Exception:
        return msgs.PoseStamped()


def get_nav_path_msgs(trs: Sequence, timestep: float, offsets=None, ctx: Context | None = None) -> list:
    """Batched get_nav_path_msg for trajectories with the same number of pieces: one evaluator launch
    per batch of equal sample counts.  `offsets` [len(trs)][3] (default zeros)."""
    ctx = ctx or default_context(7)
    trs = list(trs)
    offsets = np.zeros((len(trs), 3)) if offsets is None else np.asarray(offsets, dtype=np.float64).reshape(len(trs), 3)
    out = [None] * len(trs)
    groups = {}
    for k, tr in enumerate(trs):
        ts = np.arange(0, tr.duration, timestep)          # reference :53
        groups.setdefault((tr.n_pieces(), ts.shape[0], float(tr.duration)), []).append((k, ts))
    for (_, n_samples, _), members in groups.items():
        arrays = [trajectory_arrays(trs[k]) for k, _ in members]
        coef = np.concatenate([a[0] for a in arrays])
        dur = np.concatenate([a[1] for a in arrays])
        ts = members[0][1]
        flat = ctx.eval_flat(coef, dur, ts) if n_samples else np.empty((len(members), 0, 13))
        for row, (k, _) in enumerate(members):
            msg = _new_path("world")
            print("size:", int(trs[k].duration / timestep + 0.5))     # reference :47-48
            for s in range(n_samples):
                pose = _new_pose()
                pose.pose.position.x = float(flat[row, s, 0] + offsets[k, 0])
                pose.pose.position.y = float(flat[row, s, 1] + offsets[k, 1])
                pose.pose.position.z = float(flat[row, s, 2] + offsets[k, 2])
                q = quaternion_from_yaw(-float(flat[row, s, 12]))
                pose.pose.orientation.x, pose.pose.orientation.y = q[0], q[1]
                pose.pose.orientation.z, pose.pose.orientation.w = q[2], q[3]
                msg.poses.append(pose)
            out[k] = msg
    return out


def get_nav_path_msg(tr, timestep: float, offset=[0, 0, 0], ctx: Context | None = None):
    """Publish-ready Path of the trajectory's waypoints (reference :39-71, same arguments)."""
    msg = get_nav_path_msgs([tr], timestep, [offset], ctx)[0]
    try:
        import rospy
        rospy.loginfo("Published {} waypoints.".format(len(msg.poses)))
    except Exception:
        print("Published {} waypoints.".format(len(msg.poses)))
    return msg
