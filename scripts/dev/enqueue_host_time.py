"""Development aid: host time of one ym_batch_run_async call (no wait) against the GPU time of the call.  usage: enqueue_host_time.py [B] [loop]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
loop = len(sys.argv) > 2 and sys.argv[2] == "loop"
q, base = cfg2_scans()
m = ScanMatcher(None, loop=loop)
nq = _mk_native(q)
chains = [[_mk_native(b) for b in base] for _ in range(B)]   # distinct resident scans
pen, fine = (False, False) if loop else (True, True)
b = m.make_batch(nq, chains)
for i in range(3):
    b.run_async(pen, fine, slot=0); b.wait(0, per_chain=False)
host = []
m.profile(True)
for i in range(8):
    t = time.perf_counter(); b.run_async(pen, fine, slot=i); host.append(time.perf_counter() - t)
t = time.perf_counter()
for i in range(8):
    b.wait(i, per_chain=False)
ms, n = m.profile_read(2)
print("B %d: host %.0f us per run_async (min %.0f), GPU %.0f us per call" % (B, 1e6 * sum(host) / len(host), 1e6 * min(host), ms / n * 1e3))
