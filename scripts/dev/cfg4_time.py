"""Development aid: the cfg4 leg's enqueue pattern (distinct chains, 16 slots) with the GPU time of a call next to the wall time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
query, chains = synth.loop_batch_scans(n)
m = ScanMatcher(None, loop=True)
for ch in chains:
    for s in ch:
        s.native(0)
b = m.make_batch(query, chains)
for r in range(16):
    b.run_async(False, False, slot=r)
for r in range(16):
    b.wait(r, per_chain=False)
for rep in range(2):
    t = time.perf_counter()
    for r in range(16):
        b.run_async(False, False, slot=r)
    t1 = time.perf_counter()
    m.synchronize()
    dt = time.perf_counter() - t
    for r in range(16):
        b.wait(r, per_chain=False)
    print("16 enqueues: host %.2f ms each, wall %.2f ms each" % ((t1 - t) * 1e3 / 16, dt * 1e3 / 16))
m.profile(True)
for r in range(8):
    b.run_async(False, False, slot=r)
for r in range(8):
    b.wait(r, per_chain=False)
for w, name in enumerate(["correlate", "raster", "call"]):
    ms, k = m.profile_read(w)
    print("%s: %.1f us avg over %d" % (name, ms / max(k, 1) * 1e3, k))
