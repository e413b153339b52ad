"""Development aid: host time of a cold enqueue (every base scan re-posed): pose writes, run_async, and the GPU time beside them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C_
import numpy as np
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scene = synth.Scene()
q, base = synth.single_match_scans(scene)
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
chains = []
for c in range(n):
    rng = np.random.default_rng(100000 + c)
    chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
m = ScanMatcher()
b1 = m.make_batch(q, chains)
flat = [s_ for ch in chains for s_ in ch]
poses = np.array([(s_.corrected_pose.x, s_.corrected_pose.y, s_.corrected_pose.euler[-1]) for s_ in flat], dtype=np.float64)
hs = (C_.c_void_p * len(flat))(*[s_.native(0) for s_ in flat])
setp = m._lib.ym_scans_set_poses
dp = C_.POINTER(C_.c_double)
moved = [np.ascontiguousarray(poses + np.array([1e-4 * k, 0.0, 0.0])) for k in (1, 2)]
for i in range(3):
    setp(hs, moved[i % 2].ctypes.data_as(dp), len(flat)); b1.run_async(True, True, slot=i % 8)
for sl in range(3): b1.wait(sl, per_chain=False)
tp = tr = 0.0
m.profile(True)
t0 = time.perf_counter()
for i in range(8):
    a = time.perf_counter(); setp(hs, moved[i % 2].ctypes.data_as(dp), len(flat)); b = time.perf_counter()
    b1.run_async(True, True, slot=i % 8); c = time.perf_counter()
    tp += b - a; tr += c - b
for sl in range(8): b1.wait(sl, per_chain=False)
wall = (time.perf_counter() - t0) / 8
ms, k = m.profile_read(2)
print("per enqueue of %d chains: pose writes %.2f ms, run_async %.2f ms, wall %.2f ms, GPU %.2f ms" % (n, tp / 8 * 1e3, tr / 8 * 1e3, wall * 1e3, ms / max(k, 1)))
