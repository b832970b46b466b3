"""Minimal stand-ins for the ROS message types the two nodes exchange
(geometry_msgs/Point, Quaternion, Pose, PoseStamped, nav_msgs/Path and the
external TrajectoryPolynomialPieceMarios of scripts/drones_pols_generator.py:11-14)
so the node logic runs and is testable without a ROS installation.  With ROS
present the real message classes are used instead (same attribute names)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List


@dataclass
class Point:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0


@dataclass
class Quaternion:
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0
    w: float = 1.0


@dataclass
class Header:
    frame_id: str = ""
    stamp: float = 0.0


@dataclass
class Pose:
    position: Point = field(default_factory=Point)
    orientation: Quaternion = field(default_factory=Quaternion)


@dataclass
class PoseStamped:
    header: Header = field(default_factory=Header)
    pose: Pose = field(default_factory=Pose)


@dataclass
class Path:
    header: Header = field(default_factory=Header)
    poses: List[PoseStamped] = field(default_factory=list)


@dataclass
class TrajectoryPolynomialPieceMarios:
    cf_id: int = 0
    poly_x: List[float] = field(default_factory=list)
    poly_y: List[float] = field(default_factory=list)
    poly_z: List[float] = field(default_factory=list)
    poly_yaw: List[float] = field(default_factory=list)
    durations: List[float] = field(default_factory=list)


def path_from_arrays(positions, quats, frame_id: str = "world") -> Path:
    """positions [m,3], quats [m,4] (x,y,z,w) -> Path."""
    p = Path(header=Header(frame_id=frame_id))
    for pos, q in zip(positions, quats):
        p.poses.append(PoseStamped(header=Header(frame_id=frame_id),
                                   pose=Pose(Point(float(pos[0]), float(pos[1]), float(pos[2])),
                                             Quaternion(float(q[0]), float(q[1]), float(q[2]), float(q[3])))))
    return p


def path_to_arrays(path):
    """Path-like (ROS or stand-in) -> positions [m,3], quats [m,4] float64."""
    import numpy as np
    m = len(path.poses)
    pos = np.empty((m, 3))
    quat = np.empty((m, 4))
    for i, ps in enumerate(path.poses):
        p, q = ps.pose.position, ps.pose.orientation
        pos[i] = (p.x, p.y, p.z)
        quat[i] = (q.x, q.y, q.z, q.w)
    return pos, quat
