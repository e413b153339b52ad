"""The "match against a prebuilt map" entry on the device against what the REFERENCE's own functions
produced (tests/golden/make_golden_map.py): `occupancy_grid_map_to_correlation_grid` (/root/reference/yag_slam/helpers.py:
24-34), `find_best_pose_non_symmetric` (helpers.py:434-573) and the wrapper `match_scan_sets_with_map`
(scan_matching.py:124-173)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.util import GOLDEN, PlainScan  # noqa: E402

CASES = ["map_r05_two_scans", "map_r05_coarse_only", "map_r05_dirty_three", "map_r02_quirk"]


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    from tests.test_gpu_parity import _mk_native
    scans = [_mk_native(PlainScan(r, float(z["sensor_min_angle"]), float(z["sensor_angle_increment"]),
                                  float(z["sensor_min_range"]), float(z["sensor_range_threshold"]), p))
             for r, p in zip(z["q_ranges"], z["q_poses"])]
    cfg = dict(resolution=float(z["res"]), smear_deviation=float(z["smear"]), range_threshold=12.0)
    return z, cfg, scans


@pytest.mark.parametrize("name", CASES)
def test_map_from_occupancy_image_matches_reference(name):
    from yag_slam_amd.scan_matching import ScanMatcher
    z, cfg, _ = _load(name)
    m = ScanMatcher(cfg, semantics="yagpy")
    mp = m.correlation_grid_from_occupancy(z["image"], occupied_value=0)
    g = mp.to_numpy()
    assert g.shape == z["image"].shape
    nzy, nzx = np.nonzero(g)
    assert np.array_equal(nzy, z["cgrid_nz_y"]) and np.array_equal(nzx, z["cgrid_nz_x"])
    # the same float taps (a maximum adds no rounding); glibc's exp and numpy's may differ in the last bit
    np.testing.assert_allclose(g[nzy, nzx], z["cgrid_nz_val"], rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("how", ["occupancy", "uploaded_grid"])
def test_match_against_map_matches_reference(name, how):
    from yag_slam_amd.scan_matching import ScanMatcher
    z, cfg, scans = _load(name)
    m = ScanMatcher(cfg, semantics="yagpy")
    if how == "occupancy":
        mp = m.correlation_grid_from_occupancy(z["image"], occupied_value=0)
    else:
        g = np.zeros(z["image"].shape)
        g[z["cgrid_nz_y"], z["cgrid_nz_x"]] = z["cgrid_nz_val"]
        mp = g  # plain array: uploaded by the call, like the reference's argument
    r = m.match_scan_sets_with_map(mp, float(z["ox"]), float(z["oy"]), scans, bool(z["penalty"]), bool(z["do_fine"]))
    fin, coarse = z["final"], z["coarse"]
    assert r.meta["n_query_points"] == len(z["pts_local_x"])
    assert abs(r.meta["coarse_response"] - coarse[0]) <= 1e-12
    assert abs(r.response - float(z["response"])) <= 1e-12
    np.testing.assert_allclose(r.meta["corrected_centre"], fin[1:4], rtol=0, atol=1e-9)
    np.testing.assert_allclose(r.meta["centre"][:2], z["centre"], rtol=0, atol=0)
    cov = np.array(r.covariance)
    want = z["covariance"]
    np.testing.assert_allclose(cov, want, rtol=1e-9, atol=1e-15)
    # every query moved by the same rigid correction
    assert len(r.best_pose) == len(scans)
    from yag_slam_amd.transform import Transform
    diff = Transform(*[float(v) for v in (fin[1], fin[2])], 0.0, float(fin[3])) - Transform(float(z["centre"][0]), float(z["centre"][1]), 0.0, 0.0)
    for q, p in zip(scans, r.best_pose):
        e = q.corrected_pose + diff
        assert abs(p.x - e.x) < 1e-9 and abs(p.y - e.y) < 1e-9 and abs(p.euler[-1] - e.euler[-1]) < 1e-9


def test_map_entry_needs_the_python_semantics_and_valid_arguments():
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd._capi import YmError
    z, cfg, scans = _load("map_r05_coarse_only")
    with pytest.raises(YmError) as e:
        ScanMatcher(cfg).correlation_grid_from_occupancy(z["image"])  # Karto has no such entry
    assert e.value.code == -1 or "YAGPY" in str(e.value)
    m = ScanMatcher(cfg, semantics="yagpy")
    with pytest.raises(ValueError):
        m.upload_correlation_grid(np.zeros(5))
    mp = m.correlation_grid_from_occupancy(z["image"])
    with pytest.raises(YmError):
        m.match_scan_sets_with_map(mp, 0.0, 0.0, [], True, True)
    # an overridden coarse pass with the map's own cell size still lands on the same basin
    r = m.match_scan_sets_with_map(mp, float(z["ox"]), float(z["oy"]), scans, True, False, coarse=dict(grid_resolution=cfg["resolution"]))
    assert abs(r.response - float(z["response"])) <= 1e-12
