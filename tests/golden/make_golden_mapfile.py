#!/usr/bin/env python3
"""A yag-slam map file written by the reference's own `GraphSlam.binarize()`
(/root/reference/yag_slam/graph_slam.py:77-94, serde.py:25-95, models.py:41-53).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_mapfile.py

graph_slam.py, serde.py, models.py, graph.py, helpers.py and scan_matching.py are imported unmodified;
zlib and msgpack are the real packages.  Stand-ins created in a temp directory at run time replace
what this image lacks: `numba`, `cv2`, `sba_cpp.SPA2d`, `karto_scanmatcher` (attribute bags:
`ScanMatcherConfig` exposes the 11 keys of helpers.py:339-351 as class attributes so that serde's
`dir(ScanMatcherConfig())` lists them), and `tiny_tf.tf.Transform` (x y z qx qy qz qw, planar
compose; yaw <-> quaternion (0, 0, sin(yaw/2), cos(yaw/2)) as tiny_tf does).
Output: tests/golden/graph_ref.bin (the bytes the reference wrote) -- a data file.
"""
import math
import os
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


def _install_stubs():
    d = tempfile.mkdtemp(prefix="ymgold_stubs_")
    for m in ("numba", "karto_scanmatcher", "tiny_tf", "cv2", "sba_cpp"):
        os.makedirs(os.path.join(d, m))
    with open(os.path.join(d, "numba", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            def njit(*a, **k):
                if len(a) == 1 and callable(a[0]) and not k:
                    return a[0]
                return lambda fn: fn
            prange = range
        """))
    with open(os.path.join(d, "karto_scanmatcher", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            class ScanMatcherConfig:
                angle_variance_penalty = 0.0
                distance_variance_penalty = 0.0
                coarse_search_angle_offset = 0.0
                coarse_angle_resolution = 0.0
                fine_search_angle_resolution = 0.0
                use_response_expansion = False
                range_threshold = 0.0
                minimum_angle_penalty = 0.0
                search_size = 0.0
                resolution = 0.0
                smear_deviation = 0.0
            class Wrapper:
                def __init__(self, config): self.config = config
            class LaserScanConfig:
                def __init__(self, *a): self.args = a
            class LocalizedRangeScan:
                def __init__(self, *a): self.args = a
            class Pose2:
                def __init__(self, x=0.0, y=0.0, yaw=0.0):
                    self.x, self.y, self.yaw = x, y, yaw
            def create_occupancy_grid(*a, **k): raise NotImplementedError
        """))
    with open(os.path.join(d, "sba_cpp", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            class SPA2d:
                def __init__(self): self.nodes = []
                def add_node(self, *a): pass
                def add_constraint(self, *a): pass
                def compute(self, *a): pass
        """))
    with open(os.path.join(d, "tiny_tf", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "tiny_tf", "tf.py"), "w") as f:
        f.write(textwrap.dedent("""
            import math
            class Transform:
                def __init__(self, x=0.0, y=0.0, z=0.0, qx=0.0, qy=0.0, qz=0.0, qw=1.0):
                    self.x, self.y, self.z, self.qx, self.qy, self.qz, self.qw = x, y, z, qx, qy, qz, qw
                @classmethod
                def from_position_euler(cls, x, y, z, r, p, yaw):
                    return cls(x, y, z, 0.0, 0.0, math.sin(0.5 * yaw), math.cos(0.5 * yaw))
                @property
                def euler(self):
                    return (0.0, 0.0, 2.0 * math.atan2(self.qz, self.qw))
                def __sub__(self, o):
                    c, s = math.cos(o.euler[-1]), math.sin(o.euler[-1])
                    dx, dy = self.x - o.x, self.y - o.y
                    return Transform.from_position_euler(c * dx + s * dy, -s * dx + c * dy, 0, 0, 0, self.euler[-1] - o.euler[-1])
        """))
    with open(os.path.join(d, "cv2", "__init__.py"), "w") as f:
        f.write("")
    sys.path.insert(0, REF)
    sys.path.insert(0, d)


_install_stubs()
from yag_slam.graph_slam import GraphSlam  # noqa: E402  (the reference)
from yag_slam.models import LocalizedRangeScan  # noqa: E402  (the reference)
from yag_slam.scan_matching import Scan2DMatcherCpp  # noqa: E402  (the reference)
from yag_slam.helpers import default_config, default_config_loop  # noqa: E402
from tiny_tf.tf import Transform  # noqa: E402  (stand-in)


def main():
    rng = np.random.default_rng(2024)
    seq = Scan2DMatcherCpp(dict(default_config, search_size=0.6, smear_deviation=0.03))
    loop = Scan2DMatcherCpp(dict(default_config_loop), loop=True)
    g = GraphSlam(seq, loop, scan_buffer_len=4, loop_search_dist=2.5, loop_search_min_chain_size=3,
                  min_response_coarse=0.3, min_response_fine=0.5)
    n, beams = 9, 37
    scans = []
    for i in range(n):
        ranges = rng.uniform(0.5, 9.0, beams)
        if i == 3:
            ranges[5] = float("nan")
            ranges[6] = 31.0
        s = LocalizedRangeScan(ranges, -1.5, -1.5 + 0.0833 * (beams - 1), 0.0833, 0.05, 30.0, 12.0, 0, 0, 0)
        s.odom_pose = Transform.from_position_euler(0.3 * i + rng.normal(0, 0.02), 0.1 * i, 0, 0, 0, 0.05 * i)
        s.corrected_pose = Transform.from_position_euler(0.3 * i, 0.1 * i + 0.01, 0, 0, 0, 0.05 * i - 0.3)
        s.num = i
        scans.append(s)
        g.add_vertex(s)
        if i > 0:
            cov = (np.eye(3) * [0.01 + 0.001 * i, 0.02, 0.003] + 0.0005).tolist()
            g.link_scans(scans[i - 1], s, s.corrected_pose, cov)
    g.link_scans(scans[1], scans[7], scans[7].corrected_pose, (np.eye(3) * 0.05).tolist())
    g.running_scans = scans[-4:]
    blob = g.binarize()
    with open(os.path.join(HERE, "graph_ref.bin"), "wb") as f:
        f.write(blob)
    print("graph_ref.bin: %d bytes, %d scans, %d edges" % (len(blob), len(g.graph.vertices), len(g.graph.edges)))


if __name__ == "__main__":
    main()
