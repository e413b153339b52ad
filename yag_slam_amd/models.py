"""Scan data model: drop-in for /root/reference/yag_slam/models.py:24-116.

`LocalizedRangeScan` keeps the reference's constructor signature, pose properties with
write-through to the native twin (models.py:67-75), `points()/points_local()`, `copy()` and `num`.
The native twin `_scan` is a device-resident `ym_scan` (ranges in HBM) instead of the reference's
karto_scanmatcher C++ object (models.py:37-39); it is created on first use by a matcher so that
scans can be constructed on machines without a GPU.
"""
import ctypes as C
import math

import numpy as np

from . import _capi
from .transform import Transform


def point_readings(ranges, x, y, t, min_angle, angle_increment, range_threshold):
    """Endpoints of the readings with r <= range_threshold and not NaN, like
    `_get_point_readings` (/root/reference/yag_slam/helpers.py:58-68).  Host-side convenience for
    callers of `points()`; the matcher projects on the device."""
    r = np.asarray(ranges, dtype=np.float64)
    idx = np.nonzero(~((r > range_threshold) | np.isnan(r)))[0]
    ang = t + min_angle + idx * angle_increment
    return x + r[idx] * np.cos(ang), y + r[idx] * np.sin(ang)


def set_corrected_poses(scans, poses):
    """`scan.corrected_pose = pose` for many scans at once: what `GraphSlam.run_opt` does vertex by vertex after every
    optimisation (/root/reference/yag_slam/graph_slam.py:263-272, one pybind11 write per scan through models.py:67-75).
    poses: Transforms, or an (n, 3) array of (x, y, heading).  The Python side is updated scan by scan; the device twins
    get ONE ym_scans_set_poses call (one pose-epoch bump instead of n ctypes calls)."""
    scans = list(scans)
    if isinstance(poses, np.ndarray):
        xyz = np.ascontiguousarray(poses, dtype=np.float64).reshape(len(scans), 3)
        tfs = [Transform(float(p[0]), float(p[1]), 0.0, float(p[2])) for p in xyz]
    else:
        tfs = list(poses)
        if len(tfs) != len(scans):
            raise ValueError("set_corrected_poses: %d scans, %d poses" % (len(scans), len(tfs)))
        xyz = np.array([(float(p.x), float(p.y), float(p.euler[-1])) for p in tfs], dtype=np.float64).reshape(len(scans), 3)
    resident = [i for i, s in enumerate(scans) if getattr(s, "_native", None) is not None]
    if resident:
        handles = (C.c_void_p * len(resident))(*[scans[i]._native for i in resident])
        sub = xyz if len(resident) == len(scans) else np.ascontiguousarray(xyz[resident])
        _capi.check(_capi.lib().ym_scans_set_poses(handles, sub.ctypes.data_as(C.POINTER(C.c_double)), len(resident)))
    for s, tf in zip(scans, tfs):
        if isinstance(s, LocalizedRangeScan):
            s._corrected_pose = tf      # (the twin is already written)
        else:
            s.corrected_pose = tf       # any other scan type: its own setter


# ym_scan_desc (include/yagmatch.h) as a numpy record: descriptor arrays for ym_scans_create are filled column by column
_DESC_DTYPE = np.dtype([("ranges", "<u8"), ("n", "<i4"), ("reserved", "<i4"), ("min_angle", "<f8"), ("max_angle", "<f8"),
                        ("angle_increment", "<f8"), ("min_range", "<f8"), ("max_range", "<f8"), ("range_threshold", "<f8"),
                        ("pose", "<f8", (3,))])
assert _DESC_DTYPE.itemsize == C.sizeof(_capi.YmScanDesc)


def _create_many(device, descs):
    """ym_scans_create over a record array of descriptors -> numpy array of handles (uint64)"""
    handles = np.zeros(len(descs), dtype=np.uint64)
    if len(descs):
        _capi.check(_capi.lib().ym_scans_create(int(device), descs.ctypes.data_as(C.c_void_p), len(descs), handles.ctypes.data_as(C.c_void_p)))
    return handles


def native_many(scans, device=0):
    """Device twins for many LocalizedRangeScan objects in ONE ym_scans_create call (one pool transaction, one upload and one launch per
    2048 scans instead of a launch per scan): what `for s in scans: s.native(device)` leaves behind, bit for bit.  Scans that already
    have a twin on `device` keep it."""
    todo = [s for s in scans if not (s._native is not None and s._native_device == device)]
    if not todo:
        return
    keep = []  # (the readings must stay where the descriptors point until the call returns)
    d = np.zeros(len(todo), dtype=_DESC_DTYPE)
    for i, s in enumerate(todo):
        s._release()
        r = np.ascontiguousarray(s.ranges, dtype=np.float64)
        keep.append(r)
        p = s._corrected_pose
        d[i] = (r.ctypes.data, r.shape[0], 0, s.min_angle, s.max_angle, s.angle_increment, s.min_range, s.max_range, s.range_threshold,
                (p.x, p.y, p.euler[-1]))
    handles = _create_many(device, d)
    for s, h in zip(todo, handles):
        s._native, s._native_device = int(h), device


class ScanBlock(object):
    """n scans of ONE sensor created from arrays, without a Python object per scan: `ranges` [n][beams] float64, `poses` [n][3]
    (x, y, heading), `sensor` = (min_angle, max_angle, angle_increment, min_range, max_range, range_threshold).  For callers that
    receive thousands of scans per step (N robots, a log replayed in parallel segments): one ym_scans_create, one ym_scans_destroy.
    `.handles` (numpy uint64) are the ym_scan* the batch entry points take."""

    def __init__(self, ranges, poses, sensor, device=0):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        xyz = np.ascontiguousarray(poses, dtype=np.float64).reshape(r.shape[0], 3)
        d = np.zeros(r.shape[0], dtype=_DESC_DTYPE)
        d["ranges"] = r.ctypes.data + np.arange(r.shape[0], dtype=np.uint64) * np.uint64(r.strides[0])
        d["n"] = r.shape[1]
        for k, v in zip(("min_angle", "max_angle", "angle_increment", "min_range", "max_range", "range_threshold"), sensor):
            d[k] = v
        d["pose"] = xyz
        self.device = int(device)
        self.handles = _create_many(device, d)

    def __len__(self):
        return int(self.handles.shape[0])

    def release(self):
        if getattr(self, "handles", None) is not None and len(self.handles):
            _capi.lib().ym_scans_destroy(self.handles.ctypes.data_as(C.c_void_p), len(self.handles))
        self.handles = np.zeros(0, dtype=np.uint64)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class LocalizedRangeScan:
    def __init__(self, ranges, min_angle, max_angle, angle_increment, min_range, max_range, range_threshold,
                 x, y, t):
        self.ranges = np.array(ranges, dtype=np.float64).copy()
        self.min_angle = min_angle
        self.max_angle = max_angle
        self.angle_increment = angle_increment
        self.min_range = min_range
        self.max_range = max_range
        self.range_threshold = range_threshold

        self._odom_pose = Transform.from_position_euler(x, y, 0, 0, 0, t)
        self._corrected_pose = Transform.from_position_euler(x, y, 0, 0, 0, t)
        self._id = 0
        self._native = None      # ym_scan* (device twin), created lazily
        self._native_device = None

    # ---- native twin ---------------------------------------------------------------------
    def native(self, device=0):
        """Device-resident twin (ym_scan*), uploaded once; pose is written through on every set."""
        if self._native is not None and self._native_device == device:
            return self._native
        self._release()
        L = _capi.lib()
        d = _capi.YmScanDesc()
        r = np.ascontiguousarray(self.ranges, dtype=np.float64)
        d.ranges = r.ctypes.data_as(C.POINTER(C.c_double))
        d.n = int(r.shape[0])
        d.min_angle = float(self.min_angle)
        d.max_angle = float(self.max_angle)
        d.angle_increment = float(self.angle_increment)
        d.min_range = float(self.min_range)
        d.max_range = float(self.max_range)
        d.range_threshold = float(self.range_threshold)
        p = self._corrected_pose
        d.pose[0], d.pose[1], d.pose[2] = p.x, p.y, p.euler[-1]
        h = L.ym_scan_create(int(device), C.byref(d))
        if not h:
            raise _capi.YmError(-1, _capi.last_error())
        self._native, self._native_device = h, device
        return h

    def _release(self):
        if getattr(self, "_native", None) is not None:
            try:
                _capi.lib().ym_scan_destroy(self._native)
            except Exception:
                pass
            self._native = None

    def __del__(self):
        self._release()

    @property
    def _scan(self):
        return self.native(self._native_device or 0)

    # ---- map-file hooks: yag-slam's serde rebuilds a scan from its tagged dict (models.py:41-53) ------------
    _SENSOR_KEYS = ("min_angle", "max_angle", "angle_increment", "min_range", "max_range", "range_threshold")

    @staticmethod
    def _pose_from_tagged(d):
        """Transform from a serde dict; the `___name` class tag serde adds is not a constructor argument"""
        return Transform(**{k: v for k, v in d.items() if not k.startswith("___")})

    @classmethod
    def deserialize(cls, args):
        sensor = [args[k] for k in cls._SENSOR_KEYS]
        scan = cls(args["ranges"], *sensor, 0, 0, 0)
        scan.odom_pose = cls._pose_from_tagged(args["odom_pose"])
        scan.corrected_pose = cls._pose_from_tagged(args["corrected_pose"])
        scan.num = args["num"]
        return scan

    @classmethod
    def _deserialize(cls, **args):
        return cls.deserialize(args)

    @property
    def num(self):
        return self._id

    @num.setter
    def num(self, val):
        self._id = val

    def _get_pose(self, odom=False):
        return self._odom_pose if odom else self._corrected_pose

    def _set_pose(self, val, odom=False):
        if odom:
            self._odom_pose = val
        else:
            self._corrected_pose = val
            if self._native is not None:
                _capi.check(_capi.lib().ym_scan_set_pose(self._native, float(val.x), float(val.y),
                                                         float(val.euler[-1])))

    @property
    def odom_pose(self):
        return self._get_pose(True)

    @odom_pose.setter
    def odom_pose(self, val):
        self._set_pose(val, True)

    @property
    def corrected_pose(self):
        return self._get_pose(False)

    @corrected_pose.setter
    def corrected_pose(self, val):
        self._set_pose(val, False)

    def points(self, odom=False):
        p = self.corrected_pose if not odom else self.odom_pose
        return self.points_for_pose2d(p.x, p.y, p.euler[-1])

    def points_local(self):
        return self.points_for_pose2d(0, 0, 0)

    def points_for_pose2d(self, x, y, t):
        return point_readings(self.ranges, x, y, t, self.min_angle, self.angle_increment, self.range_threshold)

    @classmethod
    def from_json(cls, d, x, y, t, invert=True):
        """A scan from a ROS LaserScan message dumped as a dict (ranges, angle_min / _max / _increment, range_min / _max) at
        pose (x, y, t), the way /root/reference/yag_slam/models.py:110-116 reads its logs: beams reversed unless
        invert=False, the scan's own range threshold = 90 % of the sensor's maximum range."""
        beams = list(d["ranges"])
        if invert:
            beams.reverse()
        sensor = (d["angle_min"], d["angle_max"], d["angle_increment"], d["range_min"], d["range_max"], d["range_max"] * 0.9)
        return cls(beams, *sensor, x, y, t)

    def copy(self):
        """A detached scan with the same readings at the corrected pose (graph_slam.py:233 matches a moved copy);
        it gets its own device twin on first use."""
        pose = self.corrected_pose
        sensor = [getattr(self, k) for k in self._SENSOR_KEYS]
        return type(self)(self.ranges, *sensor, pose.x, pose.y, pose.euler[-1])
