"""Development aid: region correlate time of a small batch by the number of blocks that share an (item, angle block)'s regions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
scene = synth.Scene()
q, base = synth.single_match_scans(scene)
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
for B in [int(v) for v in sys.argv[1:]] or [64]:
    chains = []
    for c in range(B):
        rng = np.random.default_rng(100000 + c)
        chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    for rs in (1, 2, 3, 4, 6, 8):
        m = ScanMatcher()
        m.debug_option(28, 8)
        m.debug_option(34, rs)
        b = m.make_batch(q, chains)
        for _ in range(3):
            b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
        m.profile(True)
        for _ in range(10):
            b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
        c_ms, n = m.profile_read(0)
        r_ms, _ = m.profile_read(1)
        a_ms, _ = m.profile_read(2)
        print("B %4d rsplit %d: correlate %.1f us, raster %.1f us, call %.1f us" % (B, rs, c_ms / n * 1e3, r_ms / n * 1e3, a_ms / n * 1e3))
        m.close()
