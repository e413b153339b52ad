#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's in-tree Python matcher.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports /root/reference/yag_slam/{helpers,scan_matching}.py unmodified.  Four third-party
modules the reference imports but this image lacks are replaced by throw-away stand-ins created
in a temp directory at run time: `numba` (njit = identity, prange = range),
`karto_scanmatcher` (empty classes), `tiny_tf.tf.Transform` (x, y, euler), `cv2` (empty).  All
arithmetic that lands in the fixtures is executed by the reference's own function bodies.

Outputs: tests/golden/*.npz -- inputs (ranges, poses, sensor, config) and the reference's outputs.
The files hold data only; no reference source or bytecode is copied.
"""
import os
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

sys.dont_write_bytecode = True


def _install_stubs():
    d = tempfile.mkdtemp(prefix="ymgold_stubs_")
    os.makedirs(os.path.join(d, "numba"))
    os.makedirs(os.path.join(d, "karto_scanmatcher"))
    os.makedirs(os.path.join(d, "tiny_tf"))
    os.makedirs(os.path.join(d, "cv2"))
    with open(os.path.join(d, "numba", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            def njit(*a, **k):
                if len(a) == 1 and callable(a[0]) and not k:
                    return a[0]
                return lambda fn: fn
            prange = range
        """))
    with open(os.path.join(d, "karto_scanmatcher", "__init__.py"), "w") as f:
        f.write("class ScanMatcherConfig: pass\nclass Wrapper: pass\nclass LaserScanConfig: pass\n")
    with open(os.path.join(d, "tiny_tf", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "tiny_tf", "tf.py"), "w") as f:
        f.write(textwrap.dedent("""
            class Transform:
                def __init__(self, x, y, yaw):
                    self.x, self.y, self.euler = x, y, (0.0, 0.0, yaw)
                @classmethod
                def from_position_euler(cls, x, y, z, r, p, yaw):
                    return cls(x, y, yaw)
        """))
    with open(os.path.join(d, "cv2", "__init__.py"), "w") as f:
        f.write("")
    sys.path.insert(0, REF)
    sys.path.insert(0, d)


_install_stubs()
sys.path.insert(0, REPO)

from yag_slam import helpers as H  # noqa: E402  (the reference)
from yag_slam.scan_matching import Scan2DMatcherPy  # noqa: E402  (the reference)
from tiny_tf.tf import Transform  # noqa: E402  (stand-in)

from yag_slam_amd import synth  # noqa: E402  (own scene generator: inputs only)


class RefScan:
    """Just what scan_matching.py:187-202 touches; projection is the reference's own function."""

    def __init__(self, ranges, sensor, pose):
        self.ranges = np.array(ranges, dtype=np.float64)
        self.sensor = sensor
        self.corrected_pose = Transform(pose[0], pose[1], pose[2])

    def _pts(self, x, y, t):
        s = self.sensor
        return H._get_point_readings(self.ranges, x, y, t, s["min_angle"], 0.0,
                                     s["angle_increment"], s["range_threshold"])

    def points(self):
        p = self.corrected_pose
        return self._pts(p.x, p.y, p.euler[-1])

    def points_local(self):
        return self._pts(0, 0, 0)


def sums_volume(matcher, query, cgrid, ox, oy, coarse=True, centre=None):
    """Integer correlation sums of one find_best_pose lattice, scored by the reference's
    score_world_points_on_grid (helpers.py:149-153) on the lattice helpers.py:177-179 builds."""
    res = matcher.resolution
    if coarse:
        cx, cy, ct = query.corrected_pose.x, query.corrected_pose.y, query.corrected_pose.euler[-1]
        xy_s, xy_r, a_s, a_r = matcher.search_size * 0.5, res * 2, matcher.angle_size * 0.5, matcher.angle_res
    else:
        cx, cy, ct = centre
        xy_s, xy_r, a_s, a_r = res * 2, res, 0.0349 * 0.5, 0.00349
    xvals = np.arange(-xy_s + cx, xy_s + cx, xy_r)
    yvals = np.arange(-xy_s + cy, xy_s + cy, xy_r)
    tvals = np.arange(-a_s + ct, a_s + ct, a_r)
    ptsx, ptsy = query.points_local()
    out = np.zeros((len(tvals), len(yvals), len(xvals)), dtype=np.int64)
    for k in range(len(tvals)):
        xx, yy = H._rotate_points(ptsx, ptsy, tvals[k])
        for i in range(len(xvals)):
            for j in range(len(yvals)):
                out[k, j, i] = int(H.score_world_points_on_grid(cgrid, xvals[i] + xx, yvals[j] + yy,
                                                                ox, oy, res))
    return out, xvals, yvals, tvals


def run_case(name, sensor, cfg, base_ranges, base_poses, q_ranges, q_pose, penalty, do_fine,
             want_volume=False):
    m = Scan2DMatcherPy(cfg)
    base = [RefScan(r, sensor, p) for r, p in zip(base_ranges, base_poses)]
    query = RefScan(q_ranges, sensor, q_pose)
    r = m.match_scan(query, base, penalty, do_fine)
    grid = r.meta["grid"]
    G = grid.shape[0]
    ox = q_pose[0] - 0.5 * (G - 1) * m.resolution
    oy = q_pose[1] - 0.5 * (G - 1) * m.resolution
    nzy, nzx = np.nonzero(grid)
    out = dict(
        sensor_min_angle=sensor["min_angle"], sensor_angle_increment=sensor["angle_increment"],
        sensor_min_range=sensor["min_range"], sensor_range_threshold=sensor["range_threshold"],
        cfg_keys=np.array(sorted(cfg.keys())), cfg_vals=np.array([float(cfg[k]) for k in sorted(cfg.keys())]),
        base_ranges=np.array(base_ranges), base_poses=np.array(base_poses, dtype=np.float64),
        q_ranges=np.array(q_ranges), q_pose=np.array(q_pose, dtype=np.float64),
        penalty=int(penalty), do_fine=int(do_fine),
        response=float(r.response), covariance=np.array(r.covariance, dtype=np.float64),
        best_pose=np.array([r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1]]),
        grid_size=G, grid_nz_y=nzy.astype(np.int32), grid_nz_x=nzx.astype(np.int32),
        grid_nz_val=grid[nzy, nzx], kernel=r.meta["kernel"],
    )
    # per-base-scan valid-point survivors (helpers.py:298-329) as the matcher saw them
    kept_counts = []
    for b in base:
        px, py = b.points()
        vx, vy = H.validate_points(px, py, q_pose[0], q_pose[1])
        kept_counts.append(len(vx))
    out["valid_counts"] = np.array(kept_counts, dtype=np.int32)
    if want_volume:
        vol, xv, yv, tv = sums_volume(m, query, grid, ox, oy, coarse=True)
        out.update(coarse_sums=vol, coarse_xvals=xv, coarse_yvals=yv, coarse_tvals=tv)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s resp=%.12f pose=(%.6f, %.6f, %.6f) G=%d nz=%d -> %s" % (
        name, r.response, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1], G, len(nzx),
        os.path.relpath(path, REPO)))


def main():
    # ---- stand-alone pieces ---------------------------------------------------------------
    kern = {}
    for res, sm in [(0.01, 0.05), (0.05, 0.05), (0.05, 0.03), (0.02, 0.05), (0.005, 0.05)]:
        kern["k_%g_%g" % (res, sm)] = H.calculate_kernel(res, sm)
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **kern)

    scene = synth.Scene()
    sensor = dict(min_angle=synth.MIN_ANGLE, angle_increment=synth.ANGLE_INCREMENT,
                  min_range=synth.MIN_RANGE, range_threshold=12.0)
    base_poses, q_truth, q_prior = synth.single_match_poses()

    dirty = scene.scan_ranges(q_truth, index=50, dirty=True)
    px, py = H._get_point_readings(dirty, 3.07, 3.04, 0.05, sensor["min_angle"], 0.0,
                                   sensor["angle_increment"], sensor["range_threshold"])
    vx, vy = H.validate_points(px, py, 2.5, 2.9)
    np.savez_compressed(os.path.join(HERE, "points_dirty.npz"), ranges=dirty, pose=np.array([3.07, 3.04, 0.05]),
                        min_angle=sensor["min_angle"], angle_increment=sensor["angle_increment"],
                        range_threshold=sensor["range_threshold"], px=px, py=py,
                        viewpoint=np.array([2.5, 2.9]), vx=np.array(vx), vy=np.array(vy))
    ar = {}
    for i, (a, b, c) in enumerate([(-0.25 + 3.0, 0.25 + 3.0, 0.02), (-0.02 + 3.03, 0.02 + 3.03, 0.01),
                                   (-0.1745 + 0.05, 0.1745 + 0.05, 0.0349), (-0.02 - 7.31, 0.02 - 7.31, 0.01),
                                   (-0.15 + 1.234, 0.15 + 1.234, 0.04)]):
        ar["in_%d" % i] = np.array([a, b, c])
        ar["out_%d" % i] = np.arange(a, b, c)
    np.savez_compressed(os.path.join(HERE, "arange.npz"), **ar)

    # ---- small full matches with the complete coarse sum volume ------------------------------
    small_sensor = dict(min_angle=-1.5707963, angle_increment=0.00872664626, min_range=0.05,
                        range_threshold=6.0)
    nb = 361
    small_cfg = dict(search_size=0.3, resolution=0.02, smear_deviation=0.04,
                     coarse_search_angle_offset=0.2, coarse_angle_resolution=0.05, range_threshold=6.0)
    sb_poses = [(2.0 + 0.15 * i, 3.0, 0.02 * i) for i in range(4)]
    sb_ranges = [np.minimum(scene.scan_ranges(p, index=100 + i, n_beams=nb, min_angle=small_sensor["min_angle"],
                                              inc=small_sensor["angle_increment"]), 40.0)
                 for i, p in enumerate(sb_poses)]
    sq_truth, sq_prior = (2.66, 3.05, 0.09), (2.6, 3.0, 0.05)
    sq_ranges = scene.scan_ranges(sq_truth, index=110, n_beams=nb, min_angle=small_sensor["min_angle"],
                                  inc=small_sensor["angle_increment"])
    for pen in (0, 1):
        run_case("small_pen%d_fine1" % pen, small_sensor, small_cfg, sb_ranges, sb_poses, sq_ranges,
                 sq_prior, pen, 1, want_volume=True)
    run_case("small_pen1_fine0", small_sensor, small_cfg, sb_ranges, sb_poses, sq_ranges, sq_prior, 1, 0)
    # dirty inputs (NaN + over-range) and a rotated query
    sd_ranges = [scene.scan_ranges(p, index=120 + i, n_beams=nb, min_angle=small_sensor["min_angle"],
                                   inc=small_sensor["angle_increment"], dirty=True)
                 for i, p in enumerate(sb_poses)]
    sdq = scene.scan_ranges((2.66, 3.05, 1.29), index=130, n_beams=nb, min_angle=small_sensor["min_angle"],
                            inc=small_sensor["angle_increment"], dirty=True)
    run_case("small_dirty_rot", small_sensor, small_cfg, sd_ranges, sb_poses, sdq, (2.6, 3.0, 1.25), 1, 1,
             want_volume=True)

    # ---- the reference's own smoke input (/root/reference/test.py:29-36) through the Py path ----
    t_sensor = dict(min_angle=-1.0, angle_increment=float(np.deg2rad(0.5)), min_range=0.0, range_threshold=5.0)
    t_cfg = dict(range_threshold=5.0)
    run_case("testpy_flat", t_sensor, t_cfg, [np.array([3.0] * 230)], [(0.0, 0.0, 0.0)],
             np.array([3.0] * 230), (1.0, 0.0, 1.57), 1, 1)

    # ---- full-size cfg2 (1081 beams vs 10-scan chain, default search) -------------------------
    base_ranges = [scene.scan_ranges(p, index=i) for i, p in enumerate(base_poses)]
    q_ranges = scene.scan_ranges(q_truth, index=10)
    full_cfg = dict(range_threshold=12.0)
    run_case("cfg2_pen1_fine1", sensor, full_cfg, base_ranges, base_poses, q_ranges, q_prior, 1, 1)
    run_case("cfg2_pen0_fine1", sensor, full_cfg, base_ranges, base_poses, q_ranges, q_prior, 0, 1)
    run_case("cfg2_pen1_fine0", sensor, full_cfg, base_ranges, base_poses, q_ranges, q_prior, 1, 0)


if __name__ == "__main__":
    main()
