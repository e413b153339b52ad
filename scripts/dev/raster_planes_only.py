"""Development aid (timing only: results are wrong with option 36): the batch raster with and without the row-major window stores."""
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
for po in (0, 1, 0, 1):
    m = ScanMatcher({"use_response_expansion": False})
    m.debug_option(36, po)
    b = m.make_batch(q, chains)
    for _ in range(3):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    m.profile(True)
    for _ in range(6):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    r_ms, n = m.profile_read(1)
    f_ms, _ = m.profile_read(2)
    print("window stores %s: raster %.1f us, call %.1f us" % ("off" if po else "on ", r_ms / n * 1e3, f_ms / n * 1e3))
    m.close()
