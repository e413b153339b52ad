"""Minimal planar stand-in for `tiny_tf.tf.Transform` as yag-slam uses it.

The reference hands poses around as tiny_tf Transforms (third-party, not installed here) and only
touches `.x`, `.y`, `.euler[-1]`, `from_position_euler`, `from_pose2d`, `+` and `-`
(/root/reference/yag_slam/models.py:34-35,69; scan_matching.py:42,221; graph_slam.py:320-322).
`a + b` composes (a then b in a's frame); `a - b` is b^-1 composed with a, so that
`last.corrected_pose + (query.odom_pose - last.odom_pose)` is the odometry prior of
graph_slam.py:320-324.
"""
import math
from collections import namedtuple

Pose2 = namedtuple("Pose2", ["x", "y", "yaw"])


class Transform(object):
    __slots__ = ("x", "y", "z", "_yaw")

    def __init__(self, x=0.0, y=0.0, z=0.0, yaw=0.0):
        self.x = float(x)
        self.y = float(y)
        self.z = float(z)
        self._yaw = float(yaw)

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
