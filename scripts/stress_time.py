"""Ad-hoc timing of the stress and loop-closure configs (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher

q, base = cfg2_scans()
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
m = ScanMatcher(dict(search_size=2.0, resolution=0.005, coarse_search_angle_offset=0.785))
r = m.match_scan(nq, nb, True, True)
print("stress", r.response, r.meta)
for _ in range(3):
    m.match_scan(nq, nb, True, True)
N = 10
t = time.perf_counter()
for _ in range(N):
    m.match_scan(nq, nb, True, True)
dt = (time.perf_counter() - t) / N
print("stress sync match: %.1f us -> %.3e hyp/s" % (dt * 1e6, r.meta["hypotheses"] / dt))
m.profile(True)
for _ in range(5):
    m.match_scan(nq, nb, True, True)
for w, name in enumerate(["correlate", "raster", "call"]):
    ms, n = m.profile_read(w)
    print("  %s: %.1f us avg over %d" % (name, ms / max(n, 1) * 1e3, n))
nqp = r.meta["n_query_points"]; cd = r.meta["coarse_dims"]
ms, n = 0, 0
m.debug_stamps(True)
m.match_scan(nq, nb, True, True)
st = m.debug_stamps(False)
print("select_relax_kernel phases (us): records + states %.1f, relaxation %.1f, erase + table reset %.1f" % tuple(
    (st[b_] - st[a_]) / 100.0 for a_, b_ in ((24, 3), (3, 30), (30, 31))))
