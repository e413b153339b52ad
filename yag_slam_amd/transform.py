"""Minimal planar stand-in for `tiny_tf.tf.Transform` as yag-slam uses it.

The reference hands poses around as tiny_tf Transforms (third-party, not installed here) and only
touches `.x`, `.y`, `.euler[-1]`, `from_position_euler`, `from_pose2d`, `+` and `-`
(/root/reference/yag_slam/models.py:34-35,69; scan_matching.py:42,221; graph_slam.py:320-322).
`a + b` composes (a then b in a's frame); `a - b` is b^-1 composed with a, so that
`last.corrected_pose + (query.odom_pose - last.odom_pose)` is the odometry prior of
graph_slam.py:320-324.  yag-slam's map files store a pose as tiny_tf does, position + quaternion
(`x y z qx qy qz qw`, /root/reference/yag_slam/serde.py:94): the planar pose exposes those seven
attributes and its constructor also takes them, so serde can write and rebuild it.
"""
import math
from collections import namedtuple

Pose2 = namedtuple("Pose2", ["x", "y", "yaw"])


class Transform(object):
    __slots__ = ("x", "y", "z", "_yaw")

    def __init__(self, x=0.0, y=0.0, z=0.0, *rest, **kw):
        """Transform(x, y, z, yaw)  or, as tiny_tf spells it, Transform(x, y, z, qx, qy, qz, qw) (a rotation about z)."""
        self.x = float(x)
        self.y = float(y)
        self.z = float(z)
        quat_keys = [k for k in ("qx", "qy", "qz", "qw") if k in kw]
        if len(rest) == 4 or quat_keys:
            if len(rest) not in (0, 4) or (quat_keys and (len(rest) == 4 or "qw" not in kw)):
                raise TypeError("a quaternion needs all of qx, qy, qz, qw (got %s)" % (quat_keys or len(rest)))
            qx, qy, qz, qw = (float(v) for v in (rest if len(rest) == 4 else
                                                 (kw.get("qx", 0.0), kw.get("qy", 0.0), kw.get("qz", 0.0), kw["qw"])))
            if abs(qx) > 1e-9 or abs(qy) > 1e-9:
                # (a map file written for a 3-D pose: its yaw alone would be a wrong pose, silently)
                raise ValueError("not a planar rotation: qx = %r, qy = %r" % (qx, qy))
            self._yaw = 2.0 * math.atan2(qz, qw)
        elif len(rest) <= 1:
            self._yaw = float(rest[0] if rest else kw.get("yaw", 0.0))
        else:
            raise TypeError("Transform(x, y, z, yaw) or Transform(x, y, z, qx, qy, qz, qw)")

    qx = property(lambda self: 0.0)
    qy = property(lambda self: 0.0)
    qz = property(lambda self: math.sin(0.5 * self._yaw))
    qw = property(lambda self: math.cos(0.5 * self._yaw))

    @classmethod
    def from_position_euler(cls, x, y, z, roll, pitch, yaw):
        return cls(x, y, z, yaw)

    @classmethod
    def from_pose2d(cls, p):
        return cls(p.x, p.y, 0.0, getattr(p, "yaw", getattr(p, "t", 0.0)))

    @property
    def euler(self):
        return (0.0, 0.0, self._yaw)

    @property
    def yaw(self):
        return self._yaw

    def inverse(self):
        c, s = math.cos(self._yaw), math.sin(self._yaw)
        return Transform(-(c * self.x + s * self.y), -(-s * self.x + c * self.y), -self.z, -self._yaw)

    def __add__(self, o):
        c, s = math.cos(self._yaw), math.sin(self._yaw)
        return Transform(self.x + c * o.x - s * o.y, self.y + s * o.x + c * o.y, self.z + o.z,
                         self._yaw + o.euler[-1])

    def __sub__(self, o):
        return o.inverse() + self

    def __repr__(self):
        return "Transform(x=%r, y=%r, yaw=%r)" % (self.x, self.y, self._yaw)
