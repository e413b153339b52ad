"""Development aid: steady-state enqueue time of 4096 distinct chains (cfg2) under values of one debug option, alternating.
   AB_OPT=<option> AB_VALS=a,b,a,b python3 scripts/dev/opt_ab.py [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
scene = synth.Scene()
q, base = synth.single_match_scans(scene)
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chains = []
for c in range(B):
    rng = np.random.default_rng(100000 + c)
    chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
OPT = int(os.environ.get("AB_OPT", "31"))
VALS = [int(v) for v in os.environ.get("AB_VALS", "0,256,0,256").split(",")]
m = ScanMatcher()
b = m.make_batch(q, chains)
for v in VALS:
    m.debug_option(OPT, v)
    for _ in range(3):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    m.profile(True)
    n = 12
    t = time.perf_counter()
    for i in range(n):
        b.run_async(True, True, slot=i % 4)
        if i >= 3: b.wait((i - 3) % 4, per_chain=False)
    for i in range(n - 3, n): b.wait(i % 4, per_chain=False)
    dt = (time.perf_counter() - t) / n
    prof = [m.profile_read(w) for w in range(3)]
    m.profile(False)
    print("option %d = %d: %.1f us per enqueue; correlate %.1f raster %.1f call %.1f us" % (OPT, v, dt * 1e6, *(p[0] / max(p[1], 1) * 1e3 for p in prof)))
