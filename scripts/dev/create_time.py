"""Scan creation / destruction cost: Python twin, bare C-ABI calls, and a streaming node's step (create + match + retire)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import _capi, synth
from yag_slam_amd.mapping import SequentialMapper
from yag_slam_amd.scan_matching import ScanMatcher

m = ScanMatcher()
truth, scans = synth.trajectory_scans(2000)
t = time.perf_counter()
for s in scans:
    s.native(0)
dt = time.perf_counter() - t
print("python twin: %.1f us per scan" % (dt * 1e6 / len(scans)))
assert all(_capi.lib().ym_scan_structure_trusted(s.native(0), 0) in (0, 1) for s in scans)

L = _capi.lib()
d = _capi.YmScanDesc()
r = np.ascontiguousarray(scans[0].ranges, dtype=np.float64)
d.ranges = r.ctypes.data_as(C.POINTER(C.c_double))
d.n = int(r.shape[0])
d.min_angle, d.max_angle, d.angle_increment = scans[0].min_angle, scans[0].max_angle, scans[0].angle_increment
d.min_range, d.max_range, d.range_threshold = scans[0].min_range, scans[0].max_range, scans[0].range_threshold
for rounds in range(2):
    t = time.perf_counter()
    hs = [L.ym_scan_create(0, C.byref(d)) for _ in range(2000)]
    t1 = time.perf_counter()
    for h in hs:
        L.ym_scan_destroy(h)
    t2 = time.perf_counter()
    print("C ABI: create %.1f us, destroy %.1f us per scan" % ((t1 - t) * 1e6 / 2000, (t2 - t1) * 1e6 / 2000))

# a node that receives scans one by one: twin created when the scan arrives, retired when it leaves the running buffer
truth, scans = synth.trajectory_scans(2000)
mapper = SequentialMapper(ScanMatcher())
t = time.perf_counter()
for i, s in enumerate(scans):
    s.native(0)
    mapper.process_scan(s)
    if i >= 200:
        scans[i - 200]._release()
dt = time.perf_counter() - t
print("streaming node: %.1f us per scan (create + match + retire)" % (dt * 1e6 / len(scans)))
