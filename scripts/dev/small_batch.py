"""Development aid: GPU time of one enqueue of B distinct chains (cfg2), default correlate choice vs the direct kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
scene = synth.Scene()
q, base = synth.single_match_scans(scene)
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
def chains_of(B):
    out = []
    for c in range(B):
        rng = np.random.default_rng(100000 + c)
        out.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    return out
loop = "loop" in sys.argv
ALT = [(14, 1)]
for v in list(sys.argv):
    if v.startswith("alt="):
        ALT = [tuple(int(x) for x in kv.split(":")) for kv in v[4:].split(",")]
        sys.argv.remove(v)
pen, fine = (False, False) if loop else (True, True)
for B in [int(v) for v in sys.argv[1:] if v != "loop"] or [8, 16, 32, 64, 128, 256]:
    chains = chains_of(B)
    row = []
    for opt in (0, 1):
        m = ScanMatcher(None, loop=loop)
        if opt:
            for o_, v_ in ALT:
                m.debug_option(o_, v_)
        b = m.make_batch(q, chains)
        for _ in range(3):
            b.run_async(pen, fine, slot=0); b.wait(0, per_chain=False)
        m.profile(True)
        for _ in range(10):
            b.run_async(pen, fine, slot=0); b.wait(0, per_chain=False)
        ms, k = m.profile_read(2)
        t = time.perf_counter()
        for _ in range(10):
            b.run_async(pen, fine, slot=0); b.wait(0, per_chain=False)
        wall = (time.perf_counter() - t) / 10
        row.append((ms / k * 1e3, wall * 1e6))
        m.close()
    print("B %4d: default %.0f us GPU (%.0f wall) | alternative %.0f us GPU (%.0f wall)" % (B, row[0][0], row[0][1], row[1][0], row[1][1]))
