"""`drones_traj_generator` node: rigid-body Path -> one Path per drone.

Keeps the reference node's API (scripts/drones_traj_generator.py):
`drone_positions` (:22-25), `get_drone_positions` (:41-53), `transform(path)`
(:56-89) returning (path1, path2), `callback` (:92-98), `listener` (:101-113),
topics 'rigiBodyPath' -> 'drone1Path', 'drone2Path'.  The pose arithmetic
(p' = R(q) p_k + t, q' from R) runs on the GPU through
Context.formation_transform for any number of offsets (`transform_formation`).
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from ..context import Context, default_context
from . import msgs

drone_positions = [
    [0.5, 0, 0],
    [-0.5, 0, 0],
]

trajPub1 = None
trajPub2 = None


def get_drone_positions(drone_positions: Sequence[Sequence[float]]):
    """One PoseStamped per body-frame offset, identity orientation, frame 'rb_path'.
    (The reference's version, :41-53, appends the same object K times; this one
    returns K distinct poses, which is what it set out to do.)"""
    poses = []
    for pos in drone_positions:
        poses.append(msgs.PoseStamped(header=msgs.Header(frame_id='rb_path'),
                                      pose=msgs.Pose(msgs.Point(float(pos[0]), float(pos[1]), float(pos[2])),
                                                     msgs.Quaternion(0.0, 0.0, 0.0, 1.0))))
    return poses


def transform_formation(path, offsets, ctx: Context | None = None):
    """Rigid-body Path-like + body-frame offsets [K,3] -> list of K Paths ('world')."""
    ctx = ctx or default_context(7)
    pos, quat = msgs.path_to_arrays(path)
    rb = np.concatenate([pos, quat], axis=1)
    out = ctx.formation_transform(rb, np.asarray(offsets, dtype=np.float64))
    return [msgs.path_from_arrays(out[k, :, :3], out[k, :, 3:7], frame_id="world") for k in range(out.shape[0])]


def transform(path, inverse=False):
    """The reference's two-drone transform (:56-89): returns (path1, path2)."""
    p1, p2 = transform_formation(path, drone_positions)
    return p1, p2


def callback(path):
    print("Path received...")
    print(len(path.poses))
    drone_path1, drone_path2 = transform(path)
    if trajPub1 is not None:
        trajPub1.publish(_to_ros(drone_path1))
    if trajPub2 is not None:
        trajPub2.publish(_to_ros(drone_path2))
    return drone_path1, drone_path2


def _to_ros(path):
    """Stand-in Path -> nav_msgs/Path when ROS is importable (identity otherwise)."""
    try:
        from geometry_msgs.msg import PoseStamped
        from nav_msgs.msg import Path
    except Exception:
        return path
    out = Path()
    out.header.frame_id = path.header.frame_id
    for ps in path.poses:
        q = PoseStamped()
        q.header.frame_id = ps.header.frame_id
        q.pose.position.x, q.pose.position.y, q.pose.position.z = ps.pose.position.x, ps.pose.position.y, ps.pose.position.z
        o = ps.pose.orientation
        q.pose.orientation.x, q.pose.orientation.y, q.pose.orientation.z, q.pose.orientation.w = o.x, o.y, o.z, o.w
        out.poses.append(q)
    return out


def listener():
    """ROS entry point (reference :101-113)."""
    global trajPub1, trajPub2
    import rospy
    from nav_msgs.msg import Path
    rospy.init_node('rb_path_listener', anonymous=True)
    trajPub1 = rospy.Publisher('drone1Path', Path, queue_size=10)
    trajPub2 = rospy.Publisher('drone2Path', Path, queue_size=10)
    rospy.Subscriber('rigiBodyPath', Path, callback)
    rospy.spin()


if __name__ == '__main__':
    listener()
