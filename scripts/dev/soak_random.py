"""Development aid (GPU box): a seed sweep wider than the test suite's -- random scenes / sensors / configs (tests/test_gpu_parity.py
_random_case): the single match against the oracle (grid bytes, both sum volumes, result), and a batch of 64 - 100 sub-chains (the
region correlate from the window, tile lists, hit slots, sub-block knowledge over two calls) against single calls, bit for bit.
    python3 scripts/dev/soak_random.py [first seed] [seeds]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as T
from yag_slam_amd.scan_matching import ScanMatcher
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
big = len(sys.argv) > 3  # a third argument: batches of 130 - 600 chains (fused scoring, one-block finish)
bad = 0
for seed in range(first, first + n):
    cfg, query, base, pen, fine, rng = T._random_case(seed)
    try:
        T.compare(cfg, query, base, pen, fine)
        nq, nb = T._mk_native(query), [T._mk_native(b) for b in base]
        chains = []
        for _ in range(int(rng.integers(130, 601)) if big else int(rng.integers(64, 101))):
            kind = int(rng.integers(0, 6))
            lo = int(rng.integers(0, len(nb)))
            hi = int(rng.integers(lo, len(nb))) + 1
            ch = nb[lo:hi]
            chains.append([] if kind == 0 else ch[::-1] if kind == 1 else ch)
        m = ScanMatcher(cfg)
        for rep in range(2):  # (the second call meets the first one's window memory)
            per, best = m.match_scan_batch(nq, chains if rep == 0 else chains[::-1], pen, fine)
            order = chains if rep == 0 else chains[::-1]
            singles = [m.match_scan(nq, ch, pen, fine) for ch in order]
            for a, b in zip(per, singles):
                assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta, "batch differs from the single call"
                assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
        m.close()
    except AssertionError as e:
        bad += 1
        print("seed %d FAILED: %s | cfg %s" % (seed, str(e)[:200], cfg))
print("%d seeds from %d: %d failed" % (n, first, bad))
