"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")


class PlainPose:
    def __init__(self, x, y, t):
        self.x, self.y, self.euler = float(x), float(y), (0.0, 0.0, float(t))


class PlainScan:
    """Duck-typed LocalizedRangeScan without any device twin (oracle side of the tests)."""

    def __init__(self, ranges, min_angle, angle_increment, min_range, range_threshold, pose):
        self.ranges = np.ascontiguousarray(ranges, dtype=np.float64)
        self.min_angle = float(min_angle)
        self.angle_increment = float(angle_increment)
        self.max_angle = self.min_angle + (len(self.ranges) - 1) * self.angle_increment
        self.min_range = float(min_range)
        self.max_range = 30.0
        self.range_threshold = float(range_threshold)
        self.corrected_pose = PlainPose(*pose)


def load_case(name):
    """golden .npz -> dict with cfg dict, scans (PlainScan) and expected outputs"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = {str(k): float(v) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}
    mk = lambda r, p: PlainScan(r, float(z["sensor_min_angle"]), float(z["sensor_angle_increment"]),
                                float(z["sensor_min_range"]), float(z["sensor_range_threshold"]), p)
    base = [mk(r, p) for r, p in zip(z["base_ranges"], z["base_poses"])]
    query = mk(z["q_ranges"], z["q_pose"])
    return dict(z=z, cfg=cfg, base=base, query=query, penalty=bool(z["penalty"]), do_fine=bool(z["do_fine"]))


def cfg2_scans(range_threshold=20.0, dirty=False, n_base=10):
    """The BASELINE cfg1/cfg2 inputs as PlainScans: (query at prior, base list)."""
    from yag_slam_amd import synth
    scene = synth.Scene()
    base_poses, q_truth, q_prior = synth.single_match_poses()
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, range_threshold, p)
    base = [mk(scene.scan_ranges(p, index=i, dirty=dirty), p) for i, p in enumerate(base_poses[:n_base])]
    query = mk(scene.scan_ranges(q_truth, index=10, dirty=dirty), q_prior)
    return query, base
