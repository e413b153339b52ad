"""Development aid: where the waves of correlate_region_kernel spend their clocks, by phase (needs a library built with
-DYM_RG_PROF=1: `make -C yag_slam_amd/csrc OUT=../libyagmatch_prof.so FLAGS_EXTRA=-DYM_RG_PROF=1`, YM_LIB_PATH=.../libyagmatch_prof.so).
    python3 scripts/dev/region_phases.py [B]"""
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
m = ScanMatcher()
b = m.make_batch(q, chains)
for _ in range(3):
    b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
m.debug_stamps(True)
b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
st = m.debug_stamps(False)
names = ["gather", "barrier after the gather", "staging stores (+ wait for the loads)", "barrier after the stores",
         "the rest: setup, issue of the next region's loads, scoring"]
ph = [float(v) for v in st[10:14]]
tot = float(st[8])
ph.append(tot - sum(ph))
print("correlate_region_kernel, %d items: wave clocks by phase (sum over %d waves)" % (B, B * 3 * 8))
for n, v in zip(names, ph):
    print("  %-42s %6.1f %%   %9.0f clk per wave" % (n, 100.0 * v / max(tot, 1.0), v / (B * 3 * 8)))
print("  total %.0f clk per wave = %.1f us at 2.4 GHz" % (tot / (B * 3 * 8), tot / (B * 3 * 8) / 2400.0))
if st[9]:
    print("  shader clock the waves ran at: %.3f GHz (sum of s_memtime / sum of s_memrealtime x 100 MHz over the waves' lives)" % (tot / float(st[9]) * 0.1))
m.profile(True)
for _ in range(5):
    b.run_async(True, True, slot=0); b.wait(0, per_chain=False)
ms, n = m.profile_read(0)
print("correlate: %.1f us avg" % (ms / max(n, 1) * 1e3))
