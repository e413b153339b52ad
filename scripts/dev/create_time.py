import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
m = ScanMatcher()
truth, scans = synth.trajectory_scans(2000)
t = time.perf_counter()
for s in scans:
    s.native(0)
dt = time.perf_counter() - t
print("scan creation: %.1f us per scan" % (dt * 1e6 / len(scans)))
