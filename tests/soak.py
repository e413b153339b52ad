"""One long-lived matcher through a random sequence of calls against a matcher that forgets everything between calls (test
infrastructure: tests/test_gpu_parity.py::test_long_lived_matcher_through_random_call_sequences and scripts/dev/soak_calls.py).
Kinds of call: single matches, one-query batches of every size class (direct kernel / region correlate with regions dealt out / fused
scoring / tall tiles), multi-query (pairs) batches, resident batches run again, pose writes to queries and base scans in between, and
(round 6) batches of FRESH queries -- created in bulk from arrays (ym_scans_create), matched once through their handles, destroyed in bulk,
their blocks recycled by the pool while other calls are in flight.
What lives across calls and must stay invisible: the point cache, the pair lists kept from call to call, per-item knowledge of
window memory and of which items' planes lag behind, the sticky tile height, replayed plans."""
import numpy as np


def run(seed, ncalls, NCH=700, SIZES=(1, 1, 3, 9, 20, 48, 64, 100, 256, 300, 520, 700), verbose=True):
    """returns the number of calls whose results differed"""
    from yag_slam_amd import synth
    from yag_slam_amd.models import ScanBlock, native_many
    from yag_slam_amd.scan_matching import MatchBatch, ScanMatcher
    from yag_slam_amd.transform import Transform
    rng = np.random.default_rng(seed)
    scene = synth.Scene()
    base_poses, q_truth, q_prior = synth.single_match_poses()
    exact = [scene.cast(*p) for p in base_poses]
    chains = []
    for c in range(NCH):
        r = np.random.default_rng(50000 + c)
        chains.append([synth.resident_scan(e + r.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    queries = []
    for c in range(NCH):
        tp = (q_truth[0] + rng.uniform(-0.05, 0.05), q_truth[1] + rng.uniform(-0.05, 0.05), q_truth[2] + rng.uniform(-0.03, 0.03))
        pp = (q_prior[0] + rng.uniform(-0.02, 0.02), q_prior[1] + rng.uniform(-0.02, 0.02), q_prior[2] + rng.uniform(-0.01, 0.01))
        queries.append(synth.resident_scan(scene.scan_ranges(tp, index=300000 + c), pp))
    if seed % 2:  # (odd seeds: every scan's twin from one bulk creation; even seeds: one ym_scan_create each, on first use)
        native_many(queries + [s_ for ch in chains for s_ in ch], 0)
    sensor = (synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, synth.RANGE_THRESHOLD)
    m = ScanMatcher()
    key = lambda per: [(p.response, tuple(map(tuple, p.covariance)), p.best_pose.x, p.best_pose.y, p.best_pose.euler[-1], p.meta["hypotheses"]) for p in per]
    resident = {}
    bad = 0
    for call in range(ncalls):
        kind = int(rng.integers(0, 7))
        n = int(rng.choice([s_ for s_ in SIZES if s_ <= NCH]))
        lo = int(rng.integers(0, NCH - n + 1))
        pen, fine = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        qi = int(rng.integers(0, 4))  # (a few queries only: repeats hit the pair-list cache)
        what = ""
        ref = ScanMatcher()
        ref.debug_option(45, 0)
        try:
            if kind == 0 or n == 1:  # single match
                what = "single q%d chain %d" % (qi, lo)
                got, want = key([m.match_scan(queries[qi], chains[lo], pen, fine)]), key([ref.match_scan(queries[qi], chains[lo], pen, fine)])
            elif kind in (1, 2):  # one query, n chains
                what = "batch q%d x %d from %d" % (qi, n, lo)
                got, want = key(m.match_scan_batch(queries[qi], chains[lo:lo + n], pen, fine)[0]), key(ref.match_scan_batch(queries[qi], chains[lo:lo + n], pen, fine)[0])
            elif kind == 3:  # pairs
                what = "pairs x %d from %d" % (n, lo)
                got, want = key(m.match_pairs(queries[lo:lo + n], chains[lo:lo + n], pen, fine)), key(ref.match_pairs(queries[lo:lo + n], chains[lo:lo + n], pen, fine))
            elif kind == 4:  # a resident batch, created once per (kind, lo, n), run again and again
                pairs = bool(rng.integers(0, 2))
                k = (pairs, lo % 3, n)
                if k not in resident:
                    l2 = (lo % 3) * 7
                    resident[k] = (m.make_pairs_batch(queries[l2:l2 + n], chains[l2:l2 + n]) if pairs else m.make_batch(queries[qi], chains[l2:l2 + n]), l2, qi)
                b, l2, q2 = resident[k]
                what = "resident %s x %d from %d" % ("pairs" if pairs else "batch q%d" % q2, n, l2)
                b.run_async(pen, fine, slot=call % 8)
                got = key(b.wait(call % 8)[0])
                want = key(ref.match_pairs(queries[l2:l2 + n], chains[l2:l2 + n], pen, fine) if pairs else ref.match_scan_batch(queries[q2], chains[l2:l2 + n], pen, fine)[0])
            elif kind == 5:  # fresh queries: the readings of queries lo .. lo + n at new priors, as a block of handles
                what = "fresh pairs x %d from %d" % (n, lo)
                poses = np.array([(q_prior[0] + rng.uniform(-0.02, 0.02), q_prior[1] + rng.uniform(-0.02, 0.02), q_prior[2] + rng.uniform(-0.01, 0.01)) for _ in range(n)])
                blk = ScanBlock(np.stack([queries[lo + i].ranges for i in range(n)]), poses, sensor, device=0)
                flat = np.array([s_.native(0) for ch in chains[lo:lo + n] for s_ in ch], dtype=np.uint64)
                hb = MatchBatch.from_handles(m, blk.handles, flat, np.arange(n + 1) * 10)
                hb.run_async(pen, fine, slot=call % 8)
                got = key(hb.wait(call % 8)[0])
                hb.close()
                blk.release()
                fresh = [synth.resident_scan(queries[lo + i].ranges, poses[i]) for i in range(n)]
                want = key(ref.match_pairs(fresh, chains[lo:lo + n], pen, fine))
            else:  # pose writes: a query and a few base scans
                for s in [queries[int(rng.integers(0, 4))], queries[int(rng.integers(0, NCH))]] + [chains[int(rng.integers(0, NCH))][int(rng.integers(0, 10))] for _ in range(5)]:
                    p = s.corrected_pose
                    s.corrected_pose = Transform(p.x + rng.uniform(-0.01, 0.01), p.y + rng.uniform(-0.01, 0.01), 0.0, p.euler[-1] + rng.uniform(-0.005, 0.005))
                ref.close()
                continue
            if got != want:
                bad += 1
                first = next(i for i, (a, b_) in enumerate(zip(got, want)) if a != b_)
                print("call %d (%s, pen %d fine %d) DIFFERS at item %d: %s vs %s" % (call, what, pen, fine, first, got[first][:1], want[first][:1]))
        finally:
            ref.close()
    if verbose:
        print("seed %d: %d calls, %d differed" % (seed, ncalls, bad))
    m.close()
    return bad
