"""Development aid (timing only: results are wrong with option 36 != 0): the batch raster without some of its work.
   option 36 bits (a build with the experiment's branches in ym_k_raster.hpp): 1 no row-major window stores, 2 no stores at all,
   4 no row / column pass, 8 no cell loads"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene = synth.Scene()
q, base = synth.single_match_scans(scene)
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
chains = []
for c in range(B):
    rng = np.random.default_rng(100000 + c)
    chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
m = ScanMatcher({"use_response_expansion": False})
b = m.make_batch(q, chains)
for po in [int(v) for v in os.environ.get("VALS", "0,1,2,4,6,8,12,14,0").split(",")]:
    m.debug_option(36, po)
    for _ in range(3):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    m.profile(True)
    for _ in range(6):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    r_ms, n = m.profile_read(1)
    m.profile(False)
    print("option 36 = %2d: raster %.1f us" % (po, r_ms / n * 1e3))
