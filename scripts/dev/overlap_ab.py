"""Development aid: enqueue time of B distinct chains (cfg2) with the pair lists on the second stream (default) and on the call's stream."""
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
LOOP = bool(os.environ.get("AB_LOOP"))
PF = (False, False) if LOOP else (True, True)
for B in [int(v) for v in sys.argv[1:]] or [64, 128, 512, 4096]:
    chains = chains_of(B)
    res = []
    OPT = int(os.environ.get("AB_OPT", "29"))
    VALS = [int(v) for v in os.environ.get("AB_VALS", "1,0,1,0").split(",")]
    for overlap in VALS:
        m = ScanMatcher(None, loop=LOOP)
        m.debug_option(OPT, overlap)
        b = m.make_batch(q, chains)
        for _ in range(3):
            b.run_async(*PF, slot=0); out = b.wait(0, per_chain=True)
        n = 40 if B <= 512 else 10
        t = time.perf_counter()
        for i in range(n):
            b.run_async(*PF, slot=i % 4)
            if i >= 3: b.wait((i - 3) % 4, per_chain=False)
        for i in range(n - 3, n): b.wait(i % 4, per_chain=False)
        dt = (time.perf_counter() - t) / n
        res.append((overlap, dt * 1e6, [(r.response, r.best_pose.x) for r in out[0]][:3] if isinstance(out, tuple) else None))
    print(B, " ".join("%s:%.1fus" % (o, u) for o, u, _ in res), "same results:", all(r[2] == res[0][2] for r in res))
