"""Development aid: phase stamps (100 MHz) of block 0 of the raster / cells / score / finish kernels on a cfg2 batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = synth.Scene()
q, base = synth.single_match_scans(scene)
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
chains = []
for c in range(B):
    rng = np.random.default_rng(100000 + c)
    chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
m = ScanMatcher()
b = m.make_batch(q, chains)
for _ in range(3):
    b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
m.debug_stamps(True)
b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
st = m.debug_stamps(False)
t0 = min(v for v in st if v)
names = {4: "raster start", 5: "raster bitmap done", 6: "raster row pass done", 7: "raster end", 8: "corr start", 9: "corr end",
         10: "score start", 11: "score end", 12: "fine start", 16: "final start", 19: "final end"}
print("padded entries of item 0:", st[26])
print("raster: listed tiles %d, with a cell in reach %d, hit chunks per tile %.1f (batch %d)" % (st[27], st[28], st[29] / max(st[27], 1), B))
if len(sys.argv) > 2:
    m.debug_option(15, int(sys.argv[2]))
for i, v in enumerate(st):
    if v and i < 20:
        print("%2d %-22s %8.2f us" % (i, names.get(i, ""), (v - t0) / 100.0))
m.profile(True)
for _ in range(5):
    b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
for w, name in enumerate(["correlate", "raster", "call"]):
    ms, n = m.profile_read(w)
    print("%s: %.1f us avg" % (name, ms / max(n, 1) * 1e3))
