"""Ad-hoc timing of the hot path (development aid; bench.py is the judged benchmark)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher

q, base = cfg2_scans()
m = ScanMatcher()
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
r = m.match_scan(nq, nb, True, True)
print("result", r.response, r.best_pose, r.meta)
for _ in range(20):
    m.match_scan(nq, nb, True, True)
t = time.perf_counter()
N = 200
for _ in range(N):
    m.match_scan(nq, nb, True, True)
dt = (time.perf_counter() - t) / N
print("sync match: %.1f us  -> %.3e hyp/s" % (dt * 1e6, r.meta["hypotheses"] / dt))
# pipelined
m.synchronize()
t = time.perf_counter()
for rep in range(10):
    for s in range(32):
        m.match_scan_async(nq, nb, True, True, slot=s)
    for s in range(32):
        m.wait(s)
dt = (time.perf_counter() - t) / 320
print("pipelined match: %.1f us  -> %.3e hyp/s" % (dt * 1e6, r.meta["hypotheses"] / dt))
m.profile(True)
for _ in range(50):
    m.match_scan(nq, nb, True, True)
for w, name in enumerate(["correlate", "raster", "call"]):
    ms, n = m.profile_read(w)
    print("%s: %.2f us avg over %d" % (name, ms / max(n, 1) * 1e3, n))
m.profile(False)
m.debug_stamps(True)
m.match_scan(nq, nb, True, True)
m.match_scan(nq, nb, True, True)
st = m.debug_stamps(False)
t0 = st[0]
names = ["prep:start","prep:points","prep:trig","","rast:start","rast:scan","rast:rowpass","rast:end",
         "corr:start","corr:end","score:start","score:end","fine:start","fine:coarse","fine:cells","fine:end",
         "final:start","final:fties","prep:qend","final:end","base:points","base:nxt","base:chain","base:cells"]
for n, v in zip(names, st):
    if n: print("%-14s %8.2f us" % (n, (v - t0) / 100.0))
