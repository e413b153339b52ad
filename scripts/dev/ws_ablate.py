"""Development aid (timing only): the wave-specialised region correlate with its loaders or its gatherers switched off.
    python3 scripts/dev/ws_ablate.py [B]"""
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
for form, dbg, what in ((1, 0, "first form (every wave stages and gathers)"), (3, 0, "one block per item, sums in LDS"), (2, 0, "wave-specialised"), (2, 1, "... loaders move nothing"),
                        (2, 2, "... gatherers gather nothing"), (2, 3, "... neither")):
    m = ScanMatcher({"use_response_expansion": False})
    m.debug_option(32, form)
    m.debug_option(33, dbg)
    b = m.make_batch(q, chains)
    for _ in range(2):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    m.profile(True)
    for _ in range(5):
        b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
    ms, n = m.profile_read(0)
    print("%-50s correlate %.1f us" % (what, ms / max(n, 1) * 1e3))
    m.close()
