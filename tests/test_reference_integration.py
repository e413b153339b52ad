"""INTEGRATION.md section 1 as a test: the REFERENCE's own `GraphSlam` (/root/reference/yag_slam/graph_slam.py) drives this
package's scan and pose classes -- process_scan, loop-closure bookkeeping, run_opt, serialize / binarize / unbinarize --
without a line of yag-slam changed.  Build container only (needs /root/reference); the matcher is a stand-in (no GPU
here), everything else is the reference's code on yag_slam_amd objects."""
import math
from collections import namedtuple

import numpy as np
import pytest

from tests import refstubs

pytestmark = pytest.mark.skipif(not refstubs.available(), reason="needs /root/reference (build container)")

Result = namedtuple("Result", "response covariance best_pose meta")


class StandInMatcher(object):
    """the plugin surface GraphSlam uses: .config + match_scan(...) -> response, covariance, best_pose"""

    def __init__(self, truth, loop=False):
        from yag_slam_amd.config import make_config
        self.config = make_config(None, loop=loop)
        self.truth = truth
        self.calls = []

    def match_scan(self, query, base_scans, penalty=True, do_fine=False):
        from yag_slam_amd.transform import Transform
        self.calls.append((query.num, [b.num for b in base_scans], penalty, do_fine))
        t = self.truth[query.num]
        return Result(0.9, [[0.01, 0, 0], [0, 0.01, 0], [0, 0, 0.001]], Transform(t[0], t[1], 0.0, t[2]), {})


def _scans(n):
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    rng = np.random.default_rng(3)
    truth, scans = [], []
    for i in range(n):
        a = 2 * math.pi * i / 24.0          # two laps of a 1.5 m circle: the second lap revisits the first
        truth.append((1.5 * math.cos(a), 1.5 * math.sin(a), a + math.pi / 2))
        s = LocalizedRangeScan(rng.uniform(0.5, 9.0, 31), -1.3, 1.3, 2.6 / 30, 0.05, 30.0, 12.0, *truth[0])
        s.odom_pose = Transform(truth[i][0] + rng.normal(0, 0.01), truth[i][1] + rng.normal(0, 0.01), 0.0, truth[i][2])
        scans.append(s)
    return truth, scans


def test_reference_graphslam_runs_on_this_packages_objects():
    refstubs.install()
    from yag_slam.graph_slam import GraphSlam          # the reference
    from yag_slam_amd import mapfile
    from yag_slam_amd.transform import Transform
    n = 40
    truth, scans = _scans(n)
    seq, loop = StandInMatcher(truth), StandInMatcher(truth, loop=True)
    slam = GraphSlam(seq, loop, scan_buffer_len=5, loop_search_dist=1.0, loop_search_min_chain_size=3)
    closed = []
    for s in scans:
        res, c = slam.process_scan(s)
        if c:
            closed.append(s.num)
    # the reference's call pattern reached the plugin: sequential match with penalty + fine on the running chain
    assert seq.calls[0] == (1, [0], True, True)
    assert all(len(c[1]) <= 5 for c in seq.calls if c[2])
    assert closed and min(closed) >= 24, closed              # second lap closes against the first
    assert any(c[2:] == (False, False) for c in loop.calls)  # coarse loop stage (graph_slam.py:220)
    assert any(c[2:] == (False, True) for c in seq.calls)    # fine stage on the moved copy (graph_slam.py:233-236)
    assert slam.opt.computes == len(closed) and len(slam.opt.nodes) == n
    for s, t in zip(scans, truth):
        assert isinstance(s.corrected_pose, Transform)       # run_opt rebuilt every pose through tiny_tf.tf.Transform
        assert abs(s.corrected_pose.x - t[0]) < 1e-12 and abs(s.corrected_pose.euler[-1] - t[2]) < 1e-12
    # ---- the reference serialises this package's scans / poses / config, and reads them back
    d = slam.serialize()
    assert d["scans"][3]["___name"] == "LocalizedRangeScan" and d["scans"][3]["corrected_pose"]["___name"] == "Transform"
    assert d["seq_matcher_config"]["___name"] == "ScanMatcherConfig" and d["seq_matcher_config"]["search_size"] == 0.5
    blob = slam.binarize()
    back = GraphSlam.unbinarize(blob)                       # the reference's reader (its own scan class)
    assert len(back.graph.vertices) == n and len(back.graph.edges) == len(slam.graph.edges)
    for v, s in zip(back.graph.vertices, scans):
        assert np.array_equal(np.asarray(v.obj.ranges), s.ranges)
        assert abs(v.obj.corrected_pose.x - s.corrected_pose.x) < 1e-12
        dth = v.obj.corrected_pose.euler[-1] - s.corrected_pose.euler[-1]   # a quaternion keeps the heading modulo 2 pi
        assert abs(math.sin(dth)) < 1e-12 and math.cos(dth) > 0
    # ---- and this package's map-file reader agrees with the reference's on the same bytes
    mine = mapfile.loads(blob, matcher_factory=lambda cfg, loop: StandInMatcher(truth, loop))
    assert len(mine.scans) == n and len(mine.constraints) == len(slam.graph.edges)
    assert [s.num for s in mine.running_scans] == [s.num for s in slam.running_scans]
    for a, s in zip(mine.scans, scans):
        assert np.array_equal(a.ranges, s.ranges) and a.num == s.num
        assert abs(a.corrected_pose.x - s.corrected_pose.x) < 1e-12 and abs(a.odom_pose.y - s.odom_pose.y) < 1e-12


def test_transform_speaks_tiny_tfs_constructor_and_attributes():
    from yag_slam_amd.transform import Transform
    t = Transform(1.0, 2.0, 0.0, 0.7)
    q = Transform(t.x, t.y, t.z, t.qx, t.qy, t.qz, t.qw)      # serde's cls(*[x, y, z, qx, qy, qz, qw])
    assert abs(q.euler[-1] - 0.7) < 1e-15 and (q.x, q.y) == (1.0, 2.0)
    k = Transform(x=1.0, y=2.0, z=0.0, qx=0.0, qy=0.0, qz=t.qz, qw=t.qw)
    assert abs(k.euler[-1] - 0.7) < 1e-15
    with pytest.raises(TypeError):
        Transform(0, 0, 0, 1, 2)
