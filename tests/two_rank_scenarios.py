"""What the two rank processes of tests/test_gpu_two_ranks.py run (tests/conftest.py forks them before the session touches the GPU).
Both ranks use cuda:0 -- a real matcher each -- and talk over gloo, the records copied to the host for the collective
(yag_slam_amd/dist.py, _staged_through_host): the multi-GPU classes with two REAL matchers, on the one GPU a test box has."""
import os

import numpy as np


def _group(rank, world, port):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch, dist


def _key(r):
    return (r.response, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1], tuple(map(tuple, r.covariance)))


def sharded_loop(rank, world, port, n_chains=63):
    """ShardedLoopMatcher over real chains on the loop-closure lattice: uneven shards (63 = 32 + 31), a tie ACROSS the ranks at the top
    (the best chain's scans once more, 31 places further on: in the other rank's shard), then a single chain (rank 1's shard is empty)
    -- the winner, its payload and every local per-chain result must be the single-process batch's, bit for bit."""
    torch, dist = _group(rank, world, port)
    try:
        from yag_slam_amd import dist as ymdist
        from yag_slam_amd import synth
        from yag_slam_amd.models import native_many
        from yag_slam_amd.scan_matching import ScanMatcher
        query, chains = synth.loop_batch_scans(n_chains)
        native_many([query] + [s for ch in chains for s in ch], 0)
        b0 = ScanMatcher(None, loop=True).match_scan_batch(query, chains, False, False)[1]
        tie_at = (b0 + 31) % n_chains
        chains[tie_at] = chains[b0]
        assert (b0 < 32) != (tie_at < 32)  # (the two copies sit in different shards)
        out = {"expected_winner": min(b0, tie_at)}
        for label, sub in (("all", chains), ("one", chains[:1])):
            ref = ScanMatcher(None, loop=True)
            per_all, best = ref.match_scan_batch(query, sub, False, False)
            m = ScanMatcher(None, loop=True)
            sh = ymdist.ShardedLoopMatcher(m, query, sub, rank, world)
            assert (sh.lo, sh.hi) == ymdist.shard_range(len(sub), rank, world)
            rec = torch.zeros(ymdist.RECORD, dtype=torch.float64, device="cuda")
            win, allrec, per = sh.match(rec, False, False, slot=3)
            w = win.cpu().numpy()
            b = per_all[best]
            assert w[0] == b.response and int(w[1]) == best, (label, w, best, b.response)
            assert (w[2], w[3], w[4]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
            assert (w[5], w[6], w[7]) == (b.covariance[0][0], b.covariance[1][1], b.covariance[2][2])
            if label == "all":
                assert per_all[b0].response == per_all[tie_at].response and best == min(b0, tie_at)  # (the tie is real, the lowest id wins)
            assert (per is None) == (sh.hi == sh.lo)
            if per is not None:
                assert [_key(p) for p in per] == [_key(p) for p in per_all[sh.lo:sh.hi]]
            g = allrec.cpu().numpy()
            assert g.shape == (world, ymdist.RECORD) and ymdist.pick_best(g) >= 0
            if label == "one":
                assert g[1, 1] == -1.0  # rank 1's shard is empty
            # the rank-local constructor gives the same
            sh2 = ymdist.ShardedLoopMatcher.from_local_shard(ScanMatcher(None, loop=True), query, sub[sh.lo:sh.hi], sh.lo, len(sub), rank, world)
            rec2 = torch.zeros(ymdist.RECORD, dtype=torch.float64, device="cuda")
            win2, _, _ = sh2.match(rec2, False, False, slot=0)
            assert np.array_equal(win2.cpu().numpy(), w)
            out[label] = w.tolist()
        return out
    finally:
        dist.destroy_process_group()


def sharded_expansion(rank, world, port):
    """the record AFTER Karto's response expansion: no chain sees anything of the query, every item is re-run three times with a
    wider angle range on its rank, and the record each rank contributes is rewritten from the final results"""
    torch, dist = _group(rank, world, port)
    try:
        from yag_slam_amd import dist as ymdist
        from yag_slam_amd.models import LocalizedRangeScan
        from yag_slam_amd.scan_matching import ScanMatcher
        mk = lambda r, p: LocalizedRangeScan(r, -0.5, 0.5, 0.01, 0.05, 30.0, 20.0, *p)
        cfg = dict(search_size=0.3, range_threshold=12.0)
        chains = [[mk(np.full(101, 2.0 + 0.1 * c), (0.0, 0.1 * c, 0.0))] for c in range(5)]
        q = mk(np.full(101, 2.0), (10.0, 10.0, 0.0))
        ref = ScanMatcher(cfg)
        per_all, best = ref.match_scan_batch(q, chains, True, True)
        assert all(p.meta["expansions"] == 3 for p in per_all)
        sh = ymdist.ShardedLoopMatcher(ScanMatcher(cfg), q, chains, rank, world)
        rec = torch.zeros(ymdist.RECORD, dtype=torch.float64, device="cuda")
        win, allrec, per = sh.match(rec, True, True, slot=1)
        w = win.cpu().numpy()
        b = per_all[best]
        assert w[0] == b.response and int(w[1]) == best
        assert (w[5], w[6], w[7]) == (b.covariance[0][0], b.covariance[1][1], b.covariance[2][2])
        assert [_key(p) for p in per] == [_key(p) for p in per_all[sh.lo:sh.hi]]
        return w.tolist()
    finally:
        dist.destroy_process_group()


def angle_split(rank, world, port, stress=False):
    """AngleSplitMatcher with two real matchers: the default lattice (21 angles = 11 + 10) or BASELINE configs[4]'s stress lattice (46 =
    23 + 23, order-dependent smear): every rank returns the bits of an unsplit match_scan"""
    torch, dist = _group(rank, world, port)
    try:
        from yag_slam_amd import dist as ymdist
        from yag_slam_amd import synth
        from yag_slam_amd.models import native_many
        from yag_slam_amd.scan_matching import ScanMatcher
        cfg = dict(search_size=2.0, resolution=0.005, coarse_search_angle_offset=0.785) if stress else None
        q, base = synth.single_match_scans()
        native_many([q] + base, 0)
        want = ScanMatcher(cfg).match_scan(q, base, True, True)
        sp = ymdist.AngleSplitMatcher(ScanMatcher(cfg), rank, world)
        assert sp.k1 - sp.k0 in ((11, 10)[rank], 23) and sp.k0 == rank * sp.per
        got = [sp.match_scan(q, base, True, True) for _ in range(2)]  # (twice: the slice buffers are reused)
        for g in got:
            assert _key(g) == _key(want) and g.meta == want.meta
        return list(_key(want)[:4])
    finally:
        dist.destroy_process_group()
