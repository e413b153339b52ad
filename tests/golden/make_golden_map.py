#!/usr/bin/env python3
"""Golden vectors for the "match against a prebuilt map" entry, produced by IMPORTING the reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_map.py

Calls the reference's own `occupancy_grid_map_to_correlation_grid` (/root/reference/yag_slam/helpers.py:24-34) and
`find_best_pose_non_symmetric` (helpers.py:434-573) on synthetic inputs, and the reference's wrapper
`Scan2DMatcherPy.match_scan_sets_with_map` (/root/reference/yag_slam/scan_matching.py:124-173).  The wrapper cannot
run as shipped: it calls `find_best_pose_non_symmetric`, which scan_matching.py never imports (NameError).  Here the one
missing name is bound in the module's namespace -- the function bodies that compute the fixtures are the reference's,
unmodified.  Same throw-away stand-ins for numba / karto_scanmatcher / tiny_tf / cv2 as make_golden.py.

Output: tests/golden/map_*.npz (inputs + the reference's outputs; data only).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as MG  # noqa: E402  (installs the stand-ins, imports the reference)

H = MG.H
from yag_slam import scan_matching as SM  # noqa: E402  (the reference)
from yag_slam_amd import synth  # noqa: E402  (own scene generator: inputs only)

SM.find_best_pose_non_symmetric = H.find_best_pose_non_symmetric  # the import the reference forgot


def _planar_ops():
    """the stand-in tiny_tf Transform of make_golden.py has no arithmetic; the wrapper's last lines compose poses with
    + and - (stand-in code, not the reference's: the fixtures keep the numbers that come BEFORE it)"""
    import math
    T = MG.Transform

    def add(a, b):
        c, s = math.cos(a.euler[-1]), math.sin(a.euler[-1])
        return T(a.x + c * b.x - s * b.y, a.y + s * b.x + c * b.y, a.euler[-1] + b.euler[-1])

    def sub(a, b):
        c, s = math.cos(b.euler[-1]), math.sin(b.euler[-1])
        inv = T(-(c * b.x + s * b.y), -(-s * b.x + c * b.y), -b.euler[-1])
        return add(inv, a)
    T.__add__, T.__sub__ = add, sub


_planar_ops()


def occupancy_image(scene, res, ox, oy, w, h):
    """uint8 image, 0 = occupied (the reference's default occupied_value), 255 = free: the scene's wall segments drawn
    cell by cell"""
    im = np.full((h, w), 255, dtype=np.uint8)
    for x0, y0, x1, y1 in scene.segs:
        n = int(np.ceil(max(abs(x1 - x0), abs(y1 - y0)) / (res * 0.5))) + 1
        xs, ys = np.linspace(x0, x1, n), np.linspace(y0, y1, n)
        gx = np.round((xs - ox) / res).astype(int)
        gy = np.round((ys - oy) / res).astype(int)
        ok = (gx >= 0) & (gx < w) & (gy >= 0) & (gy < h)
        im[gy[ok], gx[ok]] = 0
    return im


def run_case(name, res, smear, n_beams, q_truth, q_priors, penalty, do_fine, dirty=False):
    scene = synth.Scene()
    ox, oy = -0.6, -0.45
    w, h = int((scene.width + 1.2) / res) + 3, int((scene.height + 0.9) / res) + 2  # non-square on purpose
    im = occupancy_image(scene, res, ox, oy, w, h)
    cgrid = H.occupancy_grid_map_to_correlation_grid(im, res, smear, 0)
    sensor = dict(min_angle=synth.MIN_ANGLE, angle_increment=synth.ANGLE_INCREMENT * (1081 - 1) / (n_beams - 1),
                  min_range=synth.MIN_RANGE, range_threshold=12.0)
    ranges = [scene.scan_ranges(t, index=600 + i, n_beams=n_beams, min_angle=sensor["min_angle"], inc=sensor["angle_increment"],
                                dirty=dirty) for i, t in enumerate(q_truth)]
    queries = [MG.RefScan(r, sensor, p) for r, p in zip(ranges, q_priors)]
    cfg = dict(resolution=res, smear_deviation=smear, range_threshold=12.0)
    m = SM.Scan2DMatcherPy(cfg)
    r = m.match_scan_sets_with_map(cgrid, ox, oy, queries, penalty, do_fine)
    # the corrected mean pose the wrapper derives its per-scan poses from: recomputed with the same two calls
    xs, ys = zip(*[(q.corrected_pose.x, q.corrected_pose.y) for q in queries])
    ox_real, oy_real = sum(xs) / float(len(xs)), sum(ys) / float(len(ys))
    xl = np.hstack([q.points()[0] for q in queries])
    yl = np.hstack([q.points()[1] for q in queries])
    xl, yl = H._transform_points(xl, yl, -ox_real, -oy_real, 0)
    c = H.find_best_pose_non_symmetric(cgrid, (xl, yl), ox_real, oy_real, 0, ox, oy, 0.25, 0.01, 0.1, 0.01, 0.05, False)
    fin = c
    if do_fine:
        fin = H.find_best_pose_non_symmetric(cgrid, (xl, yl), c[1], c[2], c[3], ox, oy, res * 2, res, 0.0349 * 0.5, 0.00349,
                                             res, penalty)
    assert float(r.response) == float(fin[0])
    nzy, nzx = np.nonzero(cgrid)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        image=im, res=res, smear=smear, ox=ox, oy=oy, penalty=int(penalty), do_fine=int(do_fine),
        sensor_min_angle=sensor["min_angle"], sensor_angle_increment=sensor["angle_increment"],
        sensor_min_range=sensor["min_range"], sensor_range_threshold=sensor["range_threshold"],
        q_ranges=np.array(ranges), q_poses=np.array(q_priors, dtype=np.float64),
        cgrid_nz_y=nzy.astype(np.int32), cgrid_nz_x=nzx.astype(np.int32), cgrid_nz_val=cgrid[nzy, nzx],
        pts_local_x=xl, pts_local_y=yl, centre=np.array([ox_real, oy_real]),
        coarse=np.array(c, dtype=np.float64), final=np.array(fin, dtype=np.float64),
        response=float(r.response), covariance=np.array(r.covariance, dtype=np.float64))
    print("%-22s %dx%d res %.3f coarse resp %.9f -> final resp %.9f pose (%.5f, %.5f, %.5f)" % (
        name, w, h, res, c[0], fin[0], fin[1], fin[2], fin[3]))


def main():
    # resolution 0.05: the cell size the reference hard-codes into the coarse pass is the map's own
    run_case("map_r05_two_scans", 0.05, 0.05, 361, [(3.07, 3.04, 0.05), (3.17, 3.05, 0.06)],
             [(3.0, 3.0, 0.0), (3.1, 3.0, 0.0)], True, True)
    run_case("map_r05_coarse_only", 0.05, 0.05, 361, [(5.2, 2.2, 1.0)], [(5.15, 2.25, 0.97)], True, False)
    run_case("map_r05_dirty_three", 0.05, 0.1, 271, [(2.0, 2.0, -0.5), (2.1, 2.0, -0.5), (2.2, 2.05, -0.45)],
             [(2.05, 1.95, -0.47), (2.15, 1.95, -0.47), (2.25, 2.0, -0.42)], False, True, dirty=True)
    # resolution 0.02: the coarse pass still indexes the map with 0.05 (scan_matching.py:153) -- reproduced as is
    run_case("map_r02_quirk", 0.02, 0.04, 181, [(3.07, 3.04, 0.05)], [(3.0, 3.0, 0.0)], True, True)


if __name__ == "__main__":
    main()
