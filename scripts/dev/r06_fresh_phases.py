#!/usr/bin/env python3
"""Where does the host time of a fresh-scan enqueue go?  One lane, one thread, 4096 new queries against 4096 resident chains per
enqueue: time of every call of the loop (creation, pairs batch, enqueue, wait, destruction), GPU otherwise idle.  YM_DEBUG_HOST=1 adds
the library's own phase times on stderr."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth  # noqa: E402
from yag_slam_amd.models import ScanBlock, native_many  # noqa: E402
from yag_slam_amd.scan_matching import MatchBatch, ScanMatcher  # noqa: E402

n = 4096
scene = synth.Scene()
base_poses, q_truth, q_prior = synth.single_match_poses()
exact = [scene.cast(*p) for p in base_poses]
chains = []
for c in range(n):
    rng = np.random.default_rng(100000 + c)
    chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
native_many([s for ch in chains for s in ch], 0)
rng = np.random.default_rng(424242)
qe = scene.cast(*q_truth)
R = np.ascontiguousarray(qe[None, :] + rng.normal(0.0, synth.SIGMA_RANGE, size=(n, qe.shape[0])))
P = np.array(q_prior)[None, :] + np.concatenate([rng.uniform(-0.02, 0.02, size=(n, 2)), rng.uniform(-0.01, 0.01, size=(n, 1))], axis=1)
sensor = (synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, synth.RANGE_THRESHOLD)
chain_h = np.array([s.native(0) for ch in chains for s in ch], dtype=np.uint64)
offs = np.arange(n + 1, dtype=np.int32) * 10
m = ScanMatcher()
acc = {}


def tick(name, t0):
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


for i in range(8):
    t0 = time.perf_counter(); blk = ScanBlock(R, P, sensor); tick("create", t0)
    t0 = time.perf_counter(); hb = MatchBatch.from_handles(m, blk.handles, chain_h, offs); tick("pairs_create", t0)
    t0 = time.perf_counter(); hb.run_async(True, True, slot=0); tick("run_async", t0)
    t0 = time.perf_counter(); hb.wait(0, per_chain=False); tick("wait", t0)
    t0 = time.perf_counter(); hb.close(); tick("close", t0)
    t0 = time.perf_counter(); blk.release(); tick("destroy", t0)
print(json.dumps({k: [round(x, 3) for x in v] for k, v in acc.items()}))
