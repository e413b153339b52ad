"""Ad-hoc timing of batched matching (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
loop = len(sys.argv) > 2 and sys.argv[2] == "loop"
q, base = cfg2_scans()
m = ScanMatcher({"use_response_expansion": False} if os.environ.get("YM_NO_EXPANSION") else None, loop=loop)
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
chains = [nb for _ in range(B)]
pen, fine = (False, False) if loop else (True, True)
per, best = m.match_scan_batch(nq, chains, pen, fine)
hyp = sum(p.meta["hypotheses"] for p in per)
print("B", B, "hyp/step", hyp, "resp0", per[0].response)
for _ in range(3):
    m.match_scan_batch(nq, chains, pen, fine)
N = 20
t = time.perf_counter()
for _ in range(N):
    m.match_scan_batch(nq, chains, pen, fine)
dt = (time.perf_counter() - t) / N
print("batch step: %.1f us -> %.3e hyp/s, %.1f us per match" % (dt * 1e6, hyp / dt, dt * 1e6 / B))
m.profile(True)
for _ in range(10):
    m.match_scan_batch(nq, chains, pen, fine)
for w, name in enumerate(["correlate", "raster", "call"]):
    ms, n = m.profile_read(w)
    print("%s: %.2f us avg over %d" % (name, ms / max(n, 1) * 1e3, n))
