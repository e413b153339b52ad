#!/usr/bin/env python3
"""Golden vectors for the HOT correlate kernels: full integer sum volumes of both find_best_pose passes, produced by the
REFERENCE's own scoring function on full-size lattices.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_sums.py

Like make_golden.py (whose stand-ins and scan class it reuses) it imports /root/reference/yag_slam/{helpers,scan_matching}.py
unmodified; every number that lands in a fixture is computed by the reference's function bodies:
    Scan2DMatcherPy.match_scan                 /root/reference/yag_slam/scan_matching.py:175-222  (result, grid)
    find_best_pose                             /root/reference/yag_slam/helpers.py:156-295        (both passes' return tuples, recorded
                                                                                                   by wrapping the name the matcher calls)
    score_world_points_on_grid + np.arange     /root/reference/yag_slam/helpers.py:134-153,177-179 (the sum volumes, make_golden.sums_volume)

Outputs tests/golden/sums_*.npz: inputs (ranges, poses, sensor, config) + coarse_sums [nt][ny][nx], fine_sums, the lattices' axes, both
passes' tuples, the final result.  Data only.  Cases: the BASELINE default configuration at full size (1081 beams, 25 x 25 x 10 -- the
lattice of the metric line), a dirty + rotated variant, the same geometry far from the origin (larger magnitudes: other roundings),
the loop-closure configuration (40 x 40 x 10 at 5 cm), a query near the range threshold.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

import make_golden as G  # noqa: E402  (installs the stand-ins, imports the reference)
import yag_slam.scan_matching as SM  # noqa: E402  (the reference)

from yag_slam_amd import synth  # noqa: E402  (own scene generator: inputs only)


def run_case(name, sensor, cfg, base_ranges, base_poses, q_ranges, q_pose, penalty, do_fine):
    m = SM.Scan2DMatcherPy(cfg)
    base = [G.RefScan(r, sensor, p) for r, p in zip(base_ranges, base_poses)]
    query = G.RefScan(q_ranges, sensor, q_pose)
    passes = []
    real = SM.find_best_pose

    def recording(*a, **k):
        out = real(*a, **k)
        passes.append([float(v) for v in out])
        return out

    SM.find_best_pose = recording
    try:
        r = m.match_scan(query, base, penalty, do_fine)
    finally:
        SM.find_best_pose = real
    grid = r.meta["grid"]
    Gs = grid.shape[0]
    ox = q_pose[0] - 0.5 * (Gs - 1) * m.resolution
    oy = q_pose[1] - 0.5 * (Gs - 1) * m.resolution
    out = dict(
        sensor_min_angle=sensor["min_angle"], sensor_angle_increment=sensor["angle_increment"],
        sensor_min_range=sensor["min_range"], sensor_range_threshold=sensor["range_threshold"],
        cfg_keys=np.array(sorted(cfg.keys())), cfg_vals=np.array([float(cfg[k]) for k in sorted(cfg.keys())]),
        base_ranges=np.array(base_ranges), base_poses=np.array(base_poses, dtype=np.float64),
        q_ranges=np.array(q_ranges), q_pose=np.array(q_pose, dtype=np.float64),
        penalty=int(penalty), do_fine=int(do_fine),
        response=float(r.response), covariance=np.array(r.covariance, dtype=np.float64),
        best_pose=np.array([r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1]]),
        grid_size=Gs, grid_nonzero=int(np.count_nonzero(grid)), grid_sum100=int(np.sum((100 * grid).astype(np.int64))),
        coarse_tuple=np.array(passes[0]),
    )
    vol, xv, yv, tv = G.sums_volume(m, query, grid, ox, oy, coarse=True)
    out.update(coarse_sums=vol.astype(np.int32), coarse_xvals=xv, coarse_yvals=yv, coarse_tvals=tv)
    if do_fine:
        out["fine_tuple"] = np.array(passes[1])
        fvol, fx, fy, ft = G.sums_volume(m, query, grid, ox, oy, coarse=False, centre=tuple(passes[0][1:4]))
        out.update(fine_sums=fvol.astype(np.int32), fine_xvals=fx, fine_yvals=fy, fine_tvals=ft)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-24s resp=%.12f pose=(%.6f, %.6f, %.6f) G=%d coarse %s fine %s" % (
        name, r.response, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1], Gs, vol.shape,
        out["fine_sums"].shape if do_fine else None))


def main():
    scene = synth.Scene()
    sensor = dict(min_angle=synth.MIN_ANGLE, angle_increment=synth.ANGLE_INCREMENT,
                  min_range=synth.MIN_RANGE, range_threshold=12.0)
    base_poses, q_truth, q_prior = synth.single_match_poses()
    base_ranges = [scene.scan_ranges(p, index=i) for i, p in enumerate(base_poses)]
    q_ranges = scene.scan_ranges(q_truth, index=10)
    full_cfg = dict(range_threshold=12.0)
    # the BASELINE default configuration at full size: the lattice of the metric line
    run_case("sums_cfg2", sensor, full_cfg, base_ranges, base_poses, q_ranges, q_prior, 1, 1)
    # dirty readings (NaN, over-range), a rotated query, mixed headings in the chain
    poses = [(4.0 + 0.08 * i, 2.6 + 0.03 * i, 2.0 + 0.05 * i) for i in range(6)]
    d_base = [scene.scan_ranges(p, index=200 + i, dirty=True) for i, p in enumerate(poses)]
    d_q = scene.scan_ranges((4.45, 2.8, 2.33), index=210, dirty=True)
    run_case("sums_dirty_rot", sensor, full_cfg, d_base, poses, d_q, (4.4, 2.75, 2.28), 0, 1)
    # the same geometry reported far from the origin (readings are pose-relative): larger magnitudes, other roundings
    shift = (-137.3, 255.1)
    far_poses = [(p[0] + shift[0], p[1] + shift[1], p[2]) for p in base_poses]
    far_q = (q_prior[0] + shift[0], q_prior[1] + shift[1], q_prior[2])
    run_case("sums_far", sensor, full_cfg, base_ranges[:5], far_poses[:5], q_ranges, far_q, 1, 1)
    # the loop-closure configuration (/root/reference/yag_slam/helpers.py:353-361): 5 cm cells, 40 x 40 x 10
    loop_cfg = dict(range_threshold=12.0, resolution=0.05, search_size=4.0, smear_deviation=0.05)
    run_case("sums_loop", sensor, loop_cfg, base_ranges[:4], base_poses[:4], q_ranges, (q_prior[0] + 0.4, q_prior[1] - 0.3, 0.1), 0, 0)
    # a matcher threshold below the scene's longest readings: the query reaches the edge of the grid
    near_sensor = dict(sensor, range_threshold=4.0)
    near_cfg = dict(range_threshold=4.0)
    run_case("sums_near_threshold", near_sensor, near_cfg, base_ranges[:3], base_poses[:3], q_ranges, q_prior, 1, 1)


if __name__ == "__main__":
    main()
