"""Occupancy-grid rendering on the device (karto_scanmatcher.create_occupancy_grid as
/root/reference/yag_slam/graph_slam.py:341-342 and /root/reference/ros1/slam_node_ros1:187-202 use it) against the CPU
oracle's sequential restatement of open_karto's OccupancyGrid.  Counts are integers: the image must be identical.
Parity unpinned at the Karto boundary (no source, no vector in the reference tree); the codes 0 / 200 / 255 are the ones
the reference's ROS node reads."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scans(n, dirty=False):
    from yag_slam_amd import synth
    scene = synth.Scene()
    truth, _ = synth.loop_trajectory(n * 12)
    poses = truth[::12]
    return scene, [synth.resident_scan(scene.scan_ranges(p, index=700 + i, dirty=dirty), p) for i, p in enumerate(poses)]


@pytest.mark.parametrize("res,rt,dirty", [(0.05, 12.0, False), (0.05, 3.0, True), (0.02, 20.0, True), (0.1, 2.5, False)])
def test_occupancy_grid_matches_oracle(res, rt, dirty):
    from oracle import oracle as orc
    from yag_slam_amd.occupancy import create_occupancy_grid
    scene, scans = _scans(40, dirty)
    g = create_occupancy_grid(scans, res, rt)
    want, (ox, oy) = orc.occupancy_grid(scans, res, rt)
    assert (g.height, g.width) == want.shape and g.image.shape == want.shape
    assert abs(g.offset.x - ox) <= 1e-12 and abs(g.offset.y - oy) <= 1e-12
    assert np.array_equal(g.image, want), int((g.image != want).sum())
    assert set(np.unique(g.image)) <= {0, 200, 255}
    if rt >= 12.0:
        # the room: walls occupied, interior free, and the grid spans what the scans saw (about 8 m x 6 m)
        assert abs(g.width * res - scene.width) < 0.3 and abs(g.height * res - scene.height) < 0.3
        free = (g.image == 255).mean()
        assert free > 0.5 and (g.image == 0).sum() > 2 * (scene.width + scene.height) / res * 0.5


def test_mapper_makes_the_occupancy_grid_like_graphslam():
    """graph_slam.py:341-342 `make_occupancy_grid(resolution, range_threshold)` on the driver, and the cleanup arithmetic of
    slam_node_ros1:190-202 applied to it (0 -> 100, 200 -> -1, 255 -> 0)"""
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import LoopClosingMapper
    from yag_slam_amd.scan_matching import ScanMatcher
    truth, scans = synth.trajectory_scans(30)
    mp = LoopClosingMapper(ScanMatcher(), None)
    for s in scans:
        mp.process_scan(s)
    g = mp.make_occupancy_grid(resolution=0.05, range_threshold=12)
    im = g.image.astype("int16")
    im[im == 0] = 100
    im[im == 200] = -1
    im[im == 255] = 0
    assert set(np.unique(im)) <= {-1, 0, 100} and (im == 100).sum() > 200 and (im == 0).sum() > (im == 100).sum()
    assert g.image.shape == (g.height, g.width) and g.resolution == 0.05
