"""BASELINE configs[2]: sequential mapping, N synthetic scans along a loop, running-chain match at each step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.mapping import SequentialMapper
from yag_slam_amd.models import LocalizedRangeScan
from yag_slam_amd.scan_matching import ScanMatcher
from yag_slam_amd.transform import Transform

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = synth.Scene()
truth, prior = synth.loop_trajectory(N)
scans = []
for i in range(N):
    r = scene.scan_ranges(truth[i], index=i)
    s = LocalizedRangeScan(r, synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE,
                           synth.RANGE_THRESHOLD, truth[0][0], truth[0][1], truth[0][2])
    # odometry = truth + drift-free noise: prior[i] is used as the odometry pose
    s.odom_pose = Transform.from_position_euler(prior[i][0], prior[i][1], 0, 0, 0, prior[i][2])
    scans.append(s)
scans[0].corrected_pose = Transform.from_position_euler(*truth[0][:2], 0, 0, 0, truth[0][2])
scans[0].odom_pose = Transform.from_position_euler(*truth[0][:2], 0, 0, 0, truth[0][2])
m = ScanMatcher()
mapper = SequentialMapper(m)
for s in scans:  # upload every scan once (resident twins)
    s.native(0)
t0 = time.perf_counter()
hyp = 0
for s in scans:
    res = mapper.process_scan(s)
    if res is not None:
        hyp += res.meta["hypotheses"]
dt = time.perf_counter() - t0
err = np.array([[s.corrected_pose.x - t[0], s.corrected_pose.y - t[1]] for s, t in zip(scans, truth)])
print("sequential mapping: %d scans in %.3f s -> %.1f matches/s, %.3e hyp/s, final position error %.3f m, max %.3f m" % (
    N, dt, (N - 1) / dt, hyp / dt, float(np.hypot(*err[-1])), float(np.hypot(err[:, 0], err[:, 1]).max())))
