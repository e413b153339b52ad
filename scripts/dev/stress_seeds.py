"""Development aid: the random cases of tests/test_gpu_parity.py (_random_case) for 90 more seeds, as batches of 8..12 chains
against single calls -- every batch kernel (tile and hit lists, region correlate with its own scoring, one-block finish)
against the single-match kernels, bit for bit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_gpu_parity import _random_case, _mk_native
from yag_slam_amd.scan_matching import ScanMatcher
bad = 0; region = 0
for seed in range(200, 290):
    cfg, query, base, pen, fine, rng = _random_case(seed)
    nq, nb = _mk_native(query), [_mk_native(b) for b in base]
    chains = []
    for _ in range(int(rng.integers(8, 13))):
        kind = int(rng.integers(0, 5)); lo = int(rng.integers(0, len(nb))); hi = int(rng.integers(lo, len(nb))) + 1
        ch = nb[lo:hi]
        chains.append([] if kind == 0 else ch[::-1] if kind == 1 else ch)
    try:
        m = ScanMatcher(cfg)
    except Exception as e:
        continue
    per, best = m.match_scan_batch(nq, chains, pen, fine)
    singles = [m.match_scan(nq, ch, pen, fine) for ch in chains]
    ok = all(a.response == b.response and a.covariance == b.covariance and a.meta == b.meta for a, b in zip(per, singles))
    d = per[0].meta["coarse_dims"]
    region += d[0] <= 26 and d[1] <= 32
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, d)
print("seeds done, mismatches:", bad, "region-eligible lattices:", region)
