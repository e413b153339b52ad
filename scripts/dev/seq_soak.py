"""Development aid: a long sequence in pieces of 1000 steps, synchronous loop and device-chained: time per step per piece."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.mapping import SequentialMapper
from yag_slam_amd.scan_matching import ScanMatcher
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
_, a = synth.trajectory_scans(N)
_, b = synth.trajectory_scans(N)
for s in a + b:
    s.native(0)
ma, mb = ScanMatcher(), ScanMatcher()
for name, scans, m, chain in (("library", a, ma, False), ("chained", b, mb, True)):
    mp = SequentialMapper(m)
    out = []
    for lo in range(0, N, 1000):
        t = time.perf_counter()
        mp.process_scans(scans[lo:lo + 1000], device_chain=chain)
        out.append((time.perf_counter() - t) * 1e6 / 1000)
    print(name, " ".join("%.1f" % v for v in out), m.sequence_stats())
pa = np.array([[s.corrected_pose.x, s.corrected_pose.y, s.corrected_pose.euler[-1]] for s in a])
pb = np.array([[s.corrected_pose.x, s.corrected_pose.y, s.corrected_pose.euler[-1]] for s in b])
print("max pose difference %.3e" % np.abs(pa - pb).max())
