"""The proof obligation behind the "yagpy" coarse pass on the production correlate kernels, checked on the CPU.

yag_lattice_kernel (yag_slam_amd/csrc/ym_k_yagpy.hpp) lets an item take its integer sums from correlate_kernel /
correlate_region_kernel / gather_kernel only if, for every (point, angle) pair, the cells the reference's Python matcher reads,

    c(i) = np.round(((xvals[i] + r) - o) / res)          /root/reference/yag_slam/helpers.py:81-83,149-153,194-196

form the lattice c(0) + 2 i.  It proves that per pair from hypothesis 0 alone: |u(0) - rint(u(0))| < 0.5 - guard with
guard = 8 (n + 8) 2^-53 M / res + 2^-40, M = |ox| + |oy| + 2 G res.  This file restates that predicate in numpy and attacks it:
readings are placed so that u(0) lands within 1e-15 .. 1e-8 cells of a rounding tie, at poses up to 10 km from the origin.
Whenever the predicate holds the lattice must hold for every i; and the attack must be real -- some of the pairs it rejects are
indeed irregular.  (No device code runs here; the device's own decision is tested in tests/test_gpu_parity.py.)"""
import numpy as np
import pytest


def _guard(nx, ox, oy, G, res):
    M = abs(ox) + abs(oy) + 2.0 * G * res
    return 8.0 * (nx + 8) * M * 2.0 ** -53 / res + 2.0 ** -40


def _lattice_ok(xv, r, o, res, step_cells):
    u = ((xv + r) - o) / res
    c = np.round(u)
    return bool(np.all(c == c[0] + step_cells * np.arange(len(xv)))), float(u[0]), float(c[0])


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("res,search,rt", [(0.01, 0.5, 12.0), (0.05, 4.0, 20.0), (0.02, 0.3, 6.0), (0.005, 2.0, 20.0)])
def test_guard_implies_lattice(seed, res, search, rt):
    rng = np.random.default_rng(1000 * seed + int(res * 1e4))
    G = int(search / res + 1 + 2 * rt / res)
    n_pass = n_reject = n_reject_irregular = 0
    for trial in range(60):
        scale = 10.0 ** rng.uniform(-1, 4)  # pose magnitudes from 0.1 m to 10 km
        cx, cy = rng.uniform(-scale, scale, size=2)
        ox, oy = cx - 0.5 * (G - 1) * res, cy - 0.5 * (G - 1) * res
        xv = np.arange(-search * 0.5 + cx, search * 0.5 + cx, res * 2)
        guard = _guard(len(xv), ox, oy, G, res)
        # cells the readings are aimed at (inside the grid), offsets from the tie from far below the guard to far above it
        cells = rng.integers(len(xv) * 2 + 2, G - len(xv) * 2 - 2, size=400)
        delta = rng.choice([-1.0, 1.0], size=400) * 10.0 ** rng.uniform(-15.5, -8, size=400)
        delta[::7] = 0.0  # exact ties too
        for m_, d in zip(cells, delta):
            r = (m_ + 0.5 + d) * res + ox - xv[0]
            ok, u0, c0 = _lattice_ok(xv, r, ox, res, 2)
            accepted = abs(u0 - c0) < 0.5 - guard
            if accepted:
                n_pass += 1
                assert ok, (res, cx, r, u0, guard)
            else:
                n_reject += 1
                n_reject_irregular += (not ok)
    assert n_reject > 0 and n_pass > 0
    assert n_reject_irregular > 0, "the attack never produced an irregular pair: it tests nothing"


def test_ordinary_readings_pass_the_guard():
    """the guard is far below the spacing of real readings: on random (not adversarial) coordinates it rejects nothing"""
    rng = np.random.default_rng(7)
    res, search, rt = 0.01, 0.5, 20.0
    G = int(search / res + 1 + 2 * rt / res)
    cx = 3.0
    ox = cx - 0.5 * (G - 1) * res
    xv = np.arange(-search * 0.5 + cx, search * 0.5 + cx, res * 2)
    guard = _guard(len(xv), ox, ox, G, res)
    r = rng.uniform(-19.0, 19.0, size=200000)
    u0 = ((xv[0] + r) - ox) / res
    assert guard < 1e-9
    assert np.all(np.abs(u0 - np.round(u0)) < 0.5 - guard)


@pytest.mark.parametrize("res", [0.01, 0.05, 0.02, 0.005, 0.25, 0.03])
def test_reciprocal_rounding_guard_implies_the_division(res):
    """yag_rint_div (ym_k_yagpy.hpp): the fine pass takes rint(d * fl(1 / res)) for rint(d / res) whenever the product lies farther than
    2^-50 |q| + 1e-12 from a rounding tie.  Attacked here with d placed from 1e-17 to 1e-9 cells off a tie, at cell numbers up to 10^9:
    whenever the guard accepts, the two roundings agree; and some of the pairs it rejects do differ (the attack is real)."""
    rng = np.random.default_rng(int(res * 1e4))
    rres = 1.0 / res
    n_accept = n_reject = n_differ = 0
    for scale in (1e1, 1e3, 1e5, 1e7, 1e9):
        cells = np.round(rng.uniform(-scale, scale, size=200000))
        delta = rng.choice([-1.0, 1.0], size=cells.size) * 10.0 ** rng.uniform(-17, -9, size=cells.size) * np.maximum(1.0, np.abs(cells) * 1e-7)
        delta[::5] = 0.0
        d = (cells + 0.5 + delta) * res
        for dd in (d, np.nextafter(d, np.inf), np.nextafter(d, -np.inf)):
            q = dd * rres
            n = np.round(q)
            accept = np.abs(q - n) < 0.5 - (np.abs(q) * 2.0 ** -50 + 1e-12)
            exact = np.round(dd / res)
            assert np.array_equal(n[accept], exact[accept])
            n_accept += int(accept.sum())
            n_reject += int((~accept).sum())
            n_differ += int((n[~accept] != exact[~accept]).sum())
    # ordinary coordinates: nothing is rejected
    d = rng.uniform(-30.0, 30.0, size=500000)
    q = d * rres
    assert np.all(np.abs(q - np.round(q)) < 0.5 - (np.abs(q) * 2.0 ** -50 + 1e-12)) or res == 0.25  # (0.25: binary fractions can tie)
    assert n_reject > 0 and n_accept > 0, (n_accept, n_reject, n_differ)
    if res != 0.25:  # (a power of two: its reciprocal is exact and the two roundings never differ)
        assert n_differ > 0, (n_accept, n_reject, n_differ)
