"""Experiment: do two matchers on two HIP streams (independent workspaces) overlap each other's latency-bound kernels?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
q, base = cfg2_scans()
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
for lanes in (1, 2, 3):
    ms = [ScanMatcher() for _ in range(lanes)]
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    for m, s in zip(ms, streams):
        m.set_stream(s.cuda_stream)
    per_lane = B // lanes
    batches = [m.make_batch(nq, [nb] * per_lane) for m in ms]
    nslots = 4
    def step(i):
        l, s = i % lanes, (i // lanes) % nslots
        if i >= nslots * lanes:
            batches[l].wait(s, per_chain=False)
        batches[l].run_async(True, True, slot=s)
    N = 120 * lanes
    W = nslots * lanes * 2
    for i in range(W): step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(W, W + N): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("lanes %d x %d items: %.1f us per %d items" % (lanes, per_lane, dt / N * lanes * 1e6, per_lane * lanes))
