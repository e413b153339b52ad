"""Pin the CPU oracle ("yagpy" semantics) against golden vectors produced by the reference's own
Python matcher (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.util import load_case, GOLDEN
import os

FULL = ["cfg2_pen1_fine1", "cfg2_pen0_fine1", "cfg2_pen1_fine0"]
SMALL = ["small_pen0_fine1", "small_pen1_fine1", "small_pen1_fine0", "small_dirty_rot"]


def test_kernels_match_reference():
    z = np.load(os.path.join(GOLDEN, "kernels.npz"))
    for key in z.files:
        _, res, sm = key.split("_")
        k = orc.kernel_yagpy(float(res), float(sm))
        assert k.shape == z[key].shape
        np.testing.assert_allclose(k, z[key], rtol=0, atol=1e-15)


def test_arange_matches_numpy():
    z = np.load(os.path.join(GOLDEN, "arange.npz"))
    for i in range(5):
        a, b, c = z["in_%d" % i]
        got = orc.arange(a, b, c)
        assert got.shape == z["out_%d" % i].shape
        assert np.array_equal(got, z["out_%d" % i])  # bit-exact, numpy's fill rule


def test_point_readings_and_validate_points():
    z = np.load(os.path.join(GOLDEN, "points_dirty.npz"))
    xs, ys = orc.point_readings(z["ranges"], float(z["min_angle"]), float(z["angle_increment"]), 0.0,
                                float(z["range_threshold"]), z["pose"], "yagpy")
    assert xs.shape == z["px"].shape
    np.testing.assert_allclose(xs, z["px"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ys, z["py"], rtol=0, atol=1e-12)
    keep = orc.valid_points(z["px"], z["py"], z["viewpoint"][0], z["viewpoint"][1], "yagpy")
    assert keep.sum() == len(z["vx"])
    assert np.array_equal(z["px"][keep], z["vx"])
    assert np.array_equal(z["py"][keep], z["vy"])
    assert not keep[0]  # reference quirk: point 0 is never kept


@pytest.mark.parametrize("name", SMALL + FULL + ["testpy_flat"])
def test_match_matches_reference(name):
    c = load_case(name)
    z = c["z"]
    o = orc.Oracle(c["cfg"], semantics="yagpy")
    r = o.match_scan(c["query"], c["base"], c["penalty"], c["do_fine"])
    # grid: identical non-zero pattern and values
    g = o.grid_f64()
    assert g.shape[0] == int(z["grid_size"])
    nzy, nzx = np.nonzero(g)
    assert np.array_equal(nzy, z["grid_nz_y"]) and np.array_equal(nzx, z["grid_nz_x"])
    np.testing.assert_allclose(g[nzy, nzx], z["grid_nz_val"], rtol=0, atol=1e-15)
    if "coarse_sums" in z.files:
        s = o.sums(0)
        assert s.shape == z["coarse_sums"].shape
        assert np.array_equal(s.astype(np.int64), z["coarse_sums"])  # integer work: bit-exact
    exp_resp = float(z["response"])
    assert abs(r["response"] - exp_resp) <= 1e-12
    np.testing.assert_allclose(r["pose"], z["best_pose"], rtol=0, atol=1e-9)
    cov = z["covariance"]
    if np.all(np.isfinite(cov)):
        np.testing.assert_allclose(r["cov"], cov, rtol=1e-9, atol=1e-15)
    else:  # response == 0 -> the reference divides by zero; same non-finite pattern expected
        assert np.array_equal(np.isfinite(r["cov"]), np.isfinite(cov))
        assert np.array_equal(np.isnan(r["cov"]), np.isnan(cov))
