#!/usr/bin/env python3
"""How long does scan creation take?  ym_scan_create one by one against ym_scans_create, array form (models.ScanBlock) and object form
(models.native_many), 4096 and 40 960 scans of 1081 beams.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth  # noqa: E402
from yag_slam_amd.models import ScanBlock, native_many  # noqa: E402

scene = synth.Scene()
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
rng = np.random.default_rng(5)
N = 40960
R = np.ascontiguousarray(np.stack([exact[i % 10] for i in range(N)]) + rng.normal(0.0, synth.SIGMA_RANGE, size=(N, exact[0].shape[0])))
P = np.array([base_poses[i % 10] for i in range(N)], dtype=np.float64)
sensor = (synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, synth.RANGE_THRESHOLD)
out = {}
blk = ScanBlock(R[:64], P[:64], sensor)  # (the pool, the staging buffers and the kernels exist)
blk.release()
for n in (4096, 40960):
    for rep in range(3):
        t0 = time.perf_counter()
        blk = ScanBlock(R[:n], P[:n], sensor)
        t1 = time.perf_counter()
        blk.release()
        t2 = time.perf_counter()
        out["array_form_%d_ms_run%d" % (n, rep)] = {"create": (t1 - t0) * 1e3, "destroy": (t2 - t1) * 1e3}
scans = [synth.resident_scan(R[i], P[i]) for i in range(4096)]
t0 = time.perf_counter()
native_many(scans, 0)
out["object_form_4096_ms"] = (time.perf_counter() - t0) * 1e3
for s in scans:
    s._release()
t0 = time.perf_counter()
for s in scans:
    s.native(0)
out["one_by_one_4096_ms"] = (time.perf_counter() - t0) * 1e3
print(json.dumps(out))
