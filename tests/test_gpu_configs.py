"""BASELINE.json configs[2] (sequential mapping, 2000 scans) and configs[3] (loop-closure batch, 4096 chains) on the
GPU, checked against the CPU oracle.  Call patterns: /root/reference/yag_slam/graph_slam.py:306-339 (process_scan)
and :217-236 (the per-chain loop of try_to_close_loop)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.util import PlainScan  # noqa: E402


def _plain(scan):
    p = scan.corrected_pose
    return PlainScan(scan.ranges, scan.min_angle, scan.angle_increment, scan.min_range, scan.range_threshold,
                     (p.x, p.y, p.euler[-1]))


class LockstepMatcher(object):
    """The matcher plugin SequentialMapper drives, with the oracle run beside every call on the SAME inputs (the scans
    at the poses the mapper holds at that moment).  Returns the GPU result, so the next step's prior is the GPU's."""

    def __init__(self, gpu, oracle, limit):
        self.gpu, self.oracle, self.limit = gpu, oracle, limit
        self.steps = 0
        self.worst = [0.0, 0.0, 0.0]

    def match_scan(self, query, base_scans, penalty=True, do_fine=False):
        r = self.gpu.match_scan(query, base_scans, penalty, do_fine)
        if self.steps < self.limit:
            ro = self.oracle.match_scan(_plain(query), [_plain(b) for b in base_scans], penalty, do_fine)
            bp = r.best_pose
            dr = abs(r.response - ro["response"])
            dp = float(np.abs(np.array([bp.x, bp.y, bp.euler[-1]]) - ro["pose"]).max())
            assert dr <= 1e-12, (self.steps, r.response, ro["response"])
            assert dp <= 1e-9, (self.steps, dp)
            np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
            assert r.meta["hypotheses"] == ro["hypotheses"] and r.meta["expansions"] == ro["expansions"]
            self.worst = [max(self.worst[0], dr), max(self.worst[1], dp), 0.0]
        self.steps += 1
        return r


def test_cfg3_sequential_mapping_lockstep_with_oracle_then_2000_scans():
    """configs[2]: 2000 scans through SequentialMapper.  The first 250 steps run the oracle in lock-step (response
    <= 1e-12, pose <= 1e-9, covariance 1e-9 relative, per step); the full run is then checked by its properties."""
    from oracle import oracle as orc
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.scan_matching import ScanMatcher
    n = 2000
    truth, scans = synth.trajectory_scans(n)
    lock = LockstepMatcher(ScanMatcher(), orc.Oracle(None, "karto"), limit=250)
    mapper = SequentialMapper(lock)
    hyp = 0
    for s in scans:
        res = mapper.process_scan(s)
        if res is not None:
            hyp += res.meta["hypotheses"]
            assert tuple(res.meta["coarse_dims"]) == (26, 26, 21) and tuple(res.meta["fine_dims"]) == (3, 3, 11)
    assert lock.steps == n - 1
    assert hyp == (n - 1) * 14295
    assert len(mapper.running_scans) == 10 and [s.num for s in mapper.running_scans] == list(range(n - 10, n))
    err = np.array([[s.corrected_pose.x - t[0], s.corrected_pose.y - t[1]] for s, t in zip(scans, truth)])
    # dead reckoning on matches against the last 10 scans only (no loop closure here): the drift over 17 laps of the
    # room stays a random walk of sub-cell steps, far below the odometry's own (0.03 m per step, uncorrelated)
    assert np.hypot(err[:, 0], err[:, 1]).max() < 0.5
    assert min(r.response for r in mapper.results) > 0.3


def test_process_scans_in_the_library_equals_the_per_scan_loop():
    """`SequentialMapper.process_scans` (ym_map_sequence: priors, matches and pose updates in one library call) against
    `process_scan` scan by scan on the same trajectory: identical bits in every pose, response and covariance; also when
    the sequence continues a running chain, and in pieces."""
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.scan_matching import ScanMatcher
    n = 120
    _, ref_scans = synth.trajectory_scans(n + 1)
    _, lib_scans = synth.trajectory_scans(n + 1)
    ref = SequentialMapper(ScanMatcher())
    ref_out = [ref.process_scan(s) for s in ref_scans[:n]]        # (ym_process_scan: one library call per scan)

    class LongWay(object):                                         # any other plugin: prior, match_scan, pose in Python
        def __init__(self):
            self.m = ScanMatcher()

        def match_scan(self, *a):
            return self.m.match_scan(*a)
    _, long_scans = synth.trajectory_scans(n + 1)
    lw = SequentialMapper(LongWay())
    long_out = [lw.process_scan(s) for s in long_scans[:n]]
    for i in range(1, n):
        a, b = ref_out[i], long_out[i]
        assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta, i
        pa, pb = ref_scans[i].corrected_pose, long_scans[i].corrected_pose
        assert (pa.x, pa.y, pa.euler[-1]) == (pb.x, pb.y, pb.euler[-1]), i
    lib = SequentialMapper(ScanMatcher())
    lib_out = lib.process_scans(lib_scans[:1]) + lib.process_scans(lib_scans[1:4])     # chain shorter than the buffer
    lib_out += [lib.process_scan(lib_scans[4])]                                        # mixed with the per-scan call
    lib_out += lib.process_scans(lib_scans[5:70]) + lib.process_scans(lib_scans[70:n])
    assert lib_out[0] is None and ref_out[0] is None and len(lib_out) == n
    for i in range(1, n):
        a, b = ref_out[i], lib_out[i]
        assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta, i
        pa, pb = ref_scans[i].corrected_pose, lib_scans[i].corrected_pose
        assert (pa.x, pa.y, pa.euler[-1]) == (pb.x, pb.y, pb.euler[-1]), i
        assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (pb.x, pb.y, pb.euler[-1]), i
        assert ref_scans[i].num == lib_scans[i].num == i
    assert [s.num for s in lib.running_scans] == [s.num for s in ref.running_scans]
    assert len(lib.results) == len(ref.results) == n - 1
    # the device twins hold the corrected poses: one more per-scan match on both mappers agrees
    ra, rb = ref.process_scan(ref_scans[n]), lib.process_scan(lib_scans[n])
    assert ra.response == rb.response and ra.covariance == rb.covariance


def test_device_chained_sequence_agrees_with_the_per_scan_loop():
    """`process_scans(device_chain=True)`: the steps enqueued back to back, each step's pose handed to the next on the
    device (the priors composed with the device's cos / sin).  Against `process_scan` scan by scan: poses and responses to
    rounding, the same lattice sizes and hypothesis counts, the same running chain; a per-scan match afterwards sees the
    poses the sequence left on the device twins."""
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.scan_matching import ScanMatcher
    n = 400
    _, ref_scans = synth.trajectory_scans(n + 1)
    _, dev_scans = synth.trajectory_scans(n + 1)
    ref = SequentialMapper(ScanMatcher())
    ref_out = [ref.process_scan(s) for s in ref_scans[:n]]
    dev = SequentialMapper(ScanMatcher())
    dev_out = dev.process_scans(dev_scans[:3], device_chain=True)          # (a chain shorter than the buffer)
    dev_out += dev.process_scans(dev_scans[3:150], device_chain=True)      # more than one segment of 128
    dev_out += [dev.process_scan(dev_scans[150])]
    dev_out += dev.process_scans(dev_scans[151:n], device_chain=True)
    assert len(dev_out) == n and dev_out[0] is None
    worst = 0.0
    for i in range(1, n):
        a, b = ref_out[i], dev_out[i]
        assert abs(a.response - b.response) <= 1e-9, (i, a.response, b.response)
        pa, pb = ref_scans[i].corrected_pose, dev_scans[i].corrected_pose
        d = max(abs(pa.x - pb.x), abs(pa.y - pb.y), abs(pa.euler[-1] - pb.euler[-1]))
        assert d <= 1e-9, (i, d)
        worst = max(worst, d)
        np.testing.assert_allclose(np.array(a.covariance), np.array(b.covariance), rtol=1e-6, atol=1e-12)
        assert a.meta["hypotheses"] == b.meta["hypotheses"] and a.meta["coarse_dims"] == b.meta["coarse_dims"]
        assert (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1]) == (pb.x, pb.y, pb.euler[-1])
    assert [s.num for s in dev.running_scans] == [s.num for s in ref.running_scans]
    ra, rb = ref.process_scan(ref_scans[n]), dev.process_scan(dev_scans[n])
    assert abs(ra.response - rb.response) <= 1e-9


def test_device_chained_sequence_recovers_from_faults():
    """What the host cannot know when it plans a chained step ahead of the device is caught on the device and the step is
    repeated synchronously: (1) the chain's cells leaving the rectangle of raster tiles the host sized from dead-reckoned
    poses -- provoked with a rectangle shrunk by debug option 25 (an odometry that turns and stretches every increment
    stays inside the real one in this room) --; (2) a scan whose odometry jumps by metres, i.e. a match that needs Karto's
    response expansion.  Results as from the per-scan loop, to rounding."""
    import math
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    n = 300

    def scans_with(odom_of):
        _, scans = synth.trajectory_scans(n)
        od = odom_of([s.odom_pose for s in scans])
        for s, o in zip(scans, od):
            s.odom_pose = o
        return scans

    def drifting(od):
        out = [od[0]]
        for a, b in zip(od, od[1:]):
            d = b - a
            out.append(out[-1] + Transform(1.06 * d.x, 1.06 * d.y, 0.0, d.euler[-1] + 0.012))
        return out

    def jumping(od):
        out = list(od)
        p = out[120]
        out[120] = Transform(p.x + 100.0, p.y - 60.0, 0.0, p.euler[-1] + 0.9)   # (one bad odometry pose: two bad increments)
        return out

    for name, odom_of, want_fault in (("drift", drifting, False), ("shrunk", drifting, True), ("jump", jumping, True)):
        ref_scans, dev_scans = scans_with(odom_of), scans_with(odom_of)
        ref = SequentialMapper(ScanMatcher())
        ref_out = ref.process_scans(ref_scans)
        dm = ScanMatcher()
        if name == "shrunk":
            dm.debug_option(25, -3)
        dev = SequentialMapper(dm)
        dev_out = dev.process_scans(dev_scans, device_chain=True)
        segments, faults, sync_steps = dm.sequence_stats()
        assert segments >= 3 and (faults >= 1) == want_fault and sync_steps == faults, (name, segments, faults, sync_steps)
        for i in range(1, n):
            a, b = ref_out[i], dev_out[i]
            assert abs(a.response - b.response) <= 1e-9, (name, i)
            assert a.meta == b.meta or a.meta["hypotheses"] == b.meta["hypotheses"], (name, i)
            pa, pb = ref_scans[i].corrected_pose, dev_scans[i].corrected_pose
            assert max(abs(pa.x - pb.x), abs(pa.y - pb.y), abs(pa.euler[-1] - pb.euler[-1])) <= 1e-9, (name, i)
        if name == "jump":
            assert max(r.meta["expansions"] for r in dev_out[1:]) >= 1
        if name == "shrunk":
            # On a matcher's first chained step every tile of the fresh window is launched, far more than the (shrunk)
            # rectangle the step may stamp; the fault has to come from THAT rectangle, or stamps outside it stay in the
            # window for good (nothing marks their tiles dirty).  One more match on the same window memory against the
            # same match on a fresh matcher: byte-identical windows.
            probe, fresh = dev_scans[n - 1].copy(), ScanMatcher()
            a, b = dm.match_scan(probe, dev.running_scans, True, True), fresh.match_scan(probe, dev.running_scans, True, True)
            assert a.response == b.response and a.covariance == b.covariance
            (ga, ia), (gb, ib) = dm.debug_grid(), fresh.debug_grid()
            assert ga.shape == gb.shape and np.array_equal(ga, gb)


def test_cfg4_loop_batch_4096_distinct_chains_against_oracle():
    """configs[3] on one GPU: the cfg2 query against 4096 distinct 10-scan chains at seeded poses (chain 0 = the query's own
    neighbourhood; every chain sees the same room, so it need not be the arg-best), loop config, penalty off, coarse only -- one match_scan_batch call.  A seeded sample of 96 chains
    (plus chain 0 and the winner) is compared with the oracle; the arg-best with numpy."""
    from oracle import oracle as orc
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    n_chains = 4096
    query, chains = synth.loop_batch_scans(n_chains)
    m = ScanMatcher(None, loop=True)
    per, best = m.match_scan_batch(query, chains, False, False)
    assert len(per) == n_chains
    resp = np.array([p.response for p in per])
    assert best == int(np.argmax(resp))
    assert resp[0] > 0.5                          # chain 0 is the query's own neighbourhood
    assert all(tuple(p.meta["coarse_dims"]) == (41, 41, 21) and p.meta["fine_dims"][0] == 0 for p in per)
    assert sum(p.meta["hypotheses"] for p in per) >= n_chains * 35301   # + response expansions, if any
    o = orc.Oracle(None, "karto", loop=True)
    rng = np.random.default_rng(4096)
    sample = sorted(set([0, best] + rng.choice(n_chains, size=96, replace=False).tolist()))
    pq = _plain(query)
    for c in sample:
        ro = o.match_scan(pq, [_plain(s) for s in chains[c]], False, False)
        r = per[c]
        assert abs(r.response - ro["response"]) <= 1e-12, (c, r.response, ro["response"])
        bp = r.best_pose
        np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], ro["pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
        assert r.meta["hypotheses"] == ro["hypotheses"] and r.meta["expansions"] == ro["expansions"]
    # the same call again on the same matcher, and as two half batches: identical bits (workspace reuse)
    per2, best2 = m.match_scan_batch(query, chains, False, False)
    assert best2 == best and all(a.response == b.response and a.covariance == b.covariance for a, b in zip(per, per2))
    half, _ = m.match_scan_batch(query, chains[2048:], False, False)
    assert all(a.response == b.response and a.covariance == b.covariance for a, b in zip(per[2048:], half))


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_loop_matcher_one_rank_rccl():
    """ShardedLoopMatcher end to end over a 1-rank RCCL group: stream-ordered record -> all-gather -> winner, against
    the plain batch path; and the record after a response expansion (rewritten by the wait)."""
    import os
    import torch
    import torch.distributed as dist
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        query, chains = synth.loop_batch_scans(24)
        m = ScanMatcher(None, loop=True)
        ref, ref_best = ScanMatcher(None, loop=True).match_scan_batch(query, chains, False, False)
        sh = ymdist.ShardedLoopMatcher(m, query, chains, 0, 1)
        rec = torch.zeros(ymdist.RECORD, dtype=torch.float64, device="cuda")
        # asynchronous form: no host wait between the enqueue and the collective
        sh.run_async(rec, False, False, slot=3)
        win, allrec = sh.reduce(rec)
        w = win.cpu().numpy()
        assert int(w[1]) == ref_best and w[0] == ref[ref_best].response
        bp = ref[ref_best].best_pose
        assert (w[2], w[3], w[4]) == (bp.x, bp.y, bp.euler[-1])
        cov = ref[ref_best].covariance
        assert (w[5], w[6], w[7]) == (cov[0][0], cov[1][1], cov[2][2])
        per, _, _ = sh.batch.wait(3)
        assert all(a.response == b.response for a, b in zip(per, ref))
        # exact form
        win2, _, per2 = sh.match(rec, False, False, slot=0)
        assert torch.equal(win2, win) and len(per2) == len(chains)
        # from a local shard with a chain id base
        sh2 = ymdist.ShardedLoopMatcher.from_local_shard(m, query, chains, 0, len(chains), 0, 1)
        win3, _, _ = sh2.match(rec, False, False, slot=1)
        assert torch.equal(win3, win)
        # response expansion: a query that sees nothing of any chain -> coarse response 0 -> retries on the host; the
        # record on the device must then be the post-expansion one
        far = synth.resident_scan(query.ranges, (40.0, 40.0, 0.0))
        sh3 = ymdist.ShardedLoopMatcher(m, far, chains[:9], 0, 1)
        win4, _, per4 = sh3.match(rec, False, False, slot=2)
        assert all(p.meta["expansions"] == 3 for p in per4)
        w4 = win4.cpu().numpy()
        assert w4[0] == max(p.response for p in per4) and int(w4[1]) == int(np.argmax([p.response for p in per4]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["default", "stress"])
def test_angle_sliced_match_equals_the_whole_match(which):
    """configs[4]'s multi-GPU split, on one GPU: the coarse angles scored in three slices (as three ranks would) into one
    response volume, per-(x, y) maxima combined with MAX, finish on the whole volume -- bit-identical to match_scan.
    Then the same through AngleSplitMatcher over a 1-rank RCCL group."""
    import os
    import torch
    import torch.distributed as dist
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    cfg = None if which == "default" else dict(search_size=2.0, resolution=0.005, coarse_search_angle_offset=0.785)
    query, base = synth.single_match_scans()
    ref = ScanMatcher(cfg).match_scan(query, base, True, True)
    m = ScanMatcher(cfg)
    nx, ny, nt = m.coarse_dims()
    assert (nx, ny, nt) == tuple(ref.meta["coarse_dims"])
    torch.cuda.set_device(0)
    st = torch.cuda.Stream()
    m.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        resp = torch.full((nt * nx * ny,), -7.0, dtype=torch.float64, device="cuda")
        cuts = [0, nt // 3, nt // 3 + 1, nt]            # uneven slices, one of a single angle
        probs = [torch.empty(nx * ny, dtype=torch.float64, device="cuda") for _ in range(3)]
        for i in range(3):
            m.slice_begin(query, base, True, True, cuts[i], cuts[i + 1], resp.data_ptr(), probs[i].data_ptr())
        probs[2].copy_(torch.maximum(torch.maximum(probs[0], probs[1]), probs[2]))
    # between slice_begin and slice_finish the matcher's synchronous slot belongs to the sliced match: anything else that
    # needs it is refused (YM_ERR_BUSY) instead of clobbering the kept plan
    from yag_slam_amd._capi import YmError
    with pytest.raises(YmError) as busy:
        m.match_scan(query, base, True, True)
    assert busy.value.code == -6
    got = m.slice_finish()
    assert got.response == ref.response and got.covariance == ref.covariance and got.meta == ref.meta
    assert (got.best_pose.x, got.best_pose.y, got.best_pose.euler[-1]) == (ref.best_pose.x, ref.best_pose.y, ref.best_pose.euler[-1])
    assert float(resp.min()) >= 0.0                    # every slice was written
    # through the class, one rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sp = ymdist.AngleSplitMatcher(ScanMatcher(cfg), 0, 1)
        for _ in range(2):
            got = sp.match_scan(query, base, True, True)
            assert got.response == ref.response and got.covariance == ref.covariance
            assert (got.best_pose.x, got.best_pose.y, got.best_pose.euler[-1]) == (ref.best_pose.x, ref.best_pose.y, ref.best_pose.euler[-1])
    finally:
        dist.destroy_process_group()



def test_cfg2_batch_512_distinct_chains_region_correlate_against_direct_and_oracle():
    """The bench's metric workload in small: the cfg2 query against 512 distinct 10-scan chains (the bench's own noisy
    copies of the base scans, seeds 100000 + chain) in ONE enqueue, penalty and refinement on.  Batches of this lattice take
    the region-staged correlate with its own scoring; every one of the 512 results must be the direct correlate's
    (option 14 = 1), bit for bit, a seeded sample of 24 chains the oracle's, and the same call in two enqueues of 256 the
    same again (workspace items reused by other chains)."""
    from oracle import oracle as orc
    from yag_slam_amd import _capi, synth
    from yag_slam_amd.scan_matching import ScanMatcher
    n_chains = 512
    scene = synth.Scene()
    q, _ = synth.single_match_scans(scene)
    base_poses, q_truth, q_prior = synth.single_match_poses()
    exact = [scene.cast(*p) for p in base_poses]
    chains = []
    for c in range(n_chains):
        rng = np.random.default_rng(100000 + c)
        chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    m = ScanMatcher()
    per, best = m.match_scan_batch(q, chains, True, True)
    assert len(per) == n_chains and all(tuple(p.meta["coarse_dims"]) == (26, 26, 21) for p in per)
    # the direct kernel; the general gather correlate; the region correlate's wave-specialised and one-block-per-item forms; sixteen
    # waves per block with regions of 80 / 100 / 128 rows (option 43)
    for opt, mode in ((14, 1), (14, 4), (32, 2), (32, 3), (43, 80), (43, 100), (43, 128)):
        md = ScanMatcher()
        try:
            if opt == 43:
                md.debug_option(32, 5)
            md.debug_option(opt, mode)
        except _capi.YmError as e:  # (option 32 = 2, 3: forms compiled only with -DYM_EXPERIMENTAL)
            assert e.code == -4 and opt in (32, 43)
            continue
        perd, bestd = md.match_scan_batch(q, chains, True, True)
        assert best == bestd == int(np.argmax([p.response for p in per]))
        for a, b in zip(per, perd):
            assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta
            assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
    o = orc.Oracle(None, "karto")
    pq = _plain(q)
    for c in sorted(np.random.default_rng(512).choice(n_chains, size=24, replace=False).tolist()):
        ro = o.match_scan(pq, [_plain(s) for s in chains[c]], True, True)
        r = per[c]
        assert abs(r.response - ro["response"]) <= 1e-12, (c, r.response, ro["response"])
        bp = r.best_pose
        np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], ro["pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
        assert r.meta["hypotheses"] == ro["hypotheses"] and r.meta["expansions"] == ro["expansions"]
    for lo in (256, 0):
        half, _ = m.match_scan_batch(q, chains[lo:lo + 256], True, True)
        assert all(a.response == b.response and a.covariance == b.covariance for a, b in zip(per[lo:lo + 256], half))


def _pairs_workload(n, scene=None):
    """n independent single-match problems along the cfg3 trajectory: query i = scan 10 + i at its odometry prior, chain i =
    the ten scans before it at their true poses (what n robots' GraphSlam.process_scan calls would send)"""
    from yag_slam_amd import synth
    truth, prior = synth.loop_trajectory(n + 10)
    scene = scene or synth.Scene()
    rng = synth.scan_ranges_many(synth.trajectory_jobs(n + 10), scene)
    queries, chains = [], []
    for i in range(n):
        queries.append(synth.resident_scan(rng[10 + i], prior[10 + i]))
        chains.append([synth.resident_scan(rng[j], truth[j]) for j in range(i, i + 10)])
    return queries, chains


@pytest.mark.parametrize("n,opts", [(64, ()), (64, ((14, 1),)), (12, ()), (64, ((14, 4),))])
def test_match_pairs_every_item_is_its_single_call(n, opts):
    """ym_match_pairs: n INDEPENDENT matches (n distinct queries at n distinct priors, each against its own chain) in one
    enqueue -- N x the call of /root/reference/yag_slam/graph_slam.py:326.  Every item must be the single call's result bit
    for bit (64 items: the region correlate, one pair list per query; option 14 = 1 the direct kernel, = 4 the gather
    correlate; 12 items: the small-batch path), a sample the oracle's; items that share a query OBJECT, a ragged and an empty
    chain ride along."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    queries, chains = _pairs_workload(n)
    # item 3 reuses item 2's query object (against its own chain), item 5 has a ragged chain, item 7 an empty one
    queries[3] = queries[2]
    chains[5] = chains[5][2:7]
    chains[7] = []
    m = ScanMatcher()
    for o_, v_ in opts:
        m.debug_option(o_, v_)
    per = m.match_pairs(queries, chains, True, True)
    assert len(per) == n
    ms = ScanMatcher()
    for i in range(n):
        s = ms.match_scan(queries[i], chains[i], True, True)
        a = per[i]
        assert a.response == s.response and a.covariance == s.covariance and a.meta == s.meta, i
        assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (s.best_pose.x, s.best_pose.y, s.best_pose.euler[-1]), i
    o = orc.Oracle(None, "karto")
    for i in sorted({2, 3, 5, 7} | set(np.random.default_rng(n).choice(n, size=6, replace=False).tolist())):
        ro = o.match_scan(_plain(queries[i]), [_plain(s) for s in chains[i]], True, True)
        r = per[i]
        assert abs(r.response - ro["response"]) <= 1e-12, (i, r.response, ro["response"])
        bp = r.best_pose
        np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], ro["pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
        assert r.meta["hypotheses"] == ro["hypotheses"] and r.meta["expansions"] == ro["expansions"]
    # the reusable form, enqueued twice with a pose write to one query and one base scan between the runs: the second run
    # sees the new poses (no stale plan), the untouched items keep their results
    b = m.make_pairs_batch(queries, chains)
    b.run_async(True, True, slot=0)
    per0, _, _ = b.wait(0)
    assert all(x.response == y.response and x.covariance == y.covariance for x, y in zip(per0, per))
    p = queries[9].corrected_pose
    queries[9].corrected_pose = type(p)(p.x + 0.02, p.y - 0.01, 0.0, p.euler[-1] + 0.01)
    p = chains[10][4].corrected_pose
    chains[10][4].corrected_pose = type(p)(p.x + 0.03, p.y, 0.0, p.euler[-1])
    b.run_async(True, True, slot=1)
    per1, _, _ = b.wait(1)
    for i in range(n):
        if i in (9, 10):
            s = ms.match_scan(queries[i], chains[i], True, True)
            assert per1[i].response == s.response and per1[i].covariance == s.covariance, i
        else:
            assert per1[i].response == per[i].response and per1[i].covariance == per[i].covariance, i


def _headline_inputs(n):
    """bench.py's metric workload (generate_inputs, rank 0's seeds): n independent cfg2 problems, resident on device 0"""
    from yag_slam_amd import synth
    from yag_slam_amd.models import native_many
    scene = synth.Scene()
    base_poses, q_truth, q_prior = synth.single_match_poses()
    exact = [scene.cast(*p) for p in base_poses]
    chains = []
    for c in range(n):
        rng = np.random.default_rng(100000 + c)
        chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    rng = np.random.default_rng(424242)
    dq_truth = np.array(q_truth) + np.concatenate([rng.uniform(-0.05, 0.05, size=(16384, 2)), rng.uniform(-0.03, 0.03, size=(16384, 1))], axis=1)
    dq_prior = np.array(q_prior) + np.concatenate([rng.uniform(-0.02, 0.02, size=(16384, 2)), rng.uniform(-0.01, 0.01, size=(16384, 1))], axis=1)
    ranges = synth.scan_ranges_many([(tuple(dq_truth[c]), 200000 + c) for c in range(n)], scene)
    queries = [synth.resident_scan(r, p) for r, p in zip(ranges, dq_prior[:n])]
    native_many(queries + [s for ch in chains for s in ch], 0)
    return queries, chains


def test_headline_workload_at_full_size():
    """The enqueue bench.py's metric line times, at its full size: 4096 INDEPENDENT cfg2 problems (every item its own 1081-beam query
    with its own readings and prior, its own 10-scan chain with its own noise; bench.py generate_inputs, rank 0's seeds), one
    ym_pairs_create batch, run twice (the second run replays the plan, as every timed step does).  EVERY item must be its single
    call's result bit for bit, and 32 seeded items the oracle's."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    n = 4096
    queries, chains = _headline_inputs(n)
    m = ScanMatcher()
    b = m.make_pairs_batch(queries, chains)
    b.run_async(True, True, slot=0)
    per, _, _ = b.wait(0)
    assert m.debug_counters()["last_correlate"] == "correlate_region_kernel"
    b.run_async(True, True, slot=1)
    again, _, _ = b.wait(1)
    for f in ("response", "pose", "cov", "hypotheses"):
        assert np.array_equal(per.array[f], again.array[f]), f
    assert len(per) == n and int(per.array["hypotheses"].min()) == 26 * 26 * 21 + 3 * 3 * 11
    ms = ScanMatcher()
    for i in range(n):
        s = ms.match_scan(queries[i], chains[i], True, True)
        r = per.array[i]
        assert s.response == r["response"] and np.array_equal(np.array(s.covariance).ravel(), r["cov"]), i
        assert (s.best_pose.x, s.best_pose.y, s.best_pose.euler[-1]) == tuple(r["pose"].tolist()), i
    o = orc.Oracle(None, "karto")
    for i in sorted(np.random.default_rng(4096).choice(n, size=32, replace=False).tolist()):
        ro = o.match_scan(_plain(queries[i]), [_plain(s) for s in chains[i]], True, True)
        r = per[i]
        assert abs(r.response - ro["response"]) <= 1e-12, (i, r.response, ro["response"])
        bp = r.best_pose
        np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], ro["pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
        assert r.meta["hypotheses"] == ro["hypotheses"] and r.meta["expansions"] == ro["expansions"]


def test_headline_workload_in_the_reference_python_semantics():
    """The same enqueue in the semantics the reference's own vectors pin ("yagpy": coarse 25 x 25 x 10 from the production region correlate for
    every item yag_lattice_kernel proves regular, fine pass by rows): 4096 independent problems in one ym_pairs_create batch.  EVERY item must
    be the pair-by-pair kernel's single call bit for bit (the Python rule as written, option 46 = 0), and 16 seeded items the oracle's --
    results and both integer sum volumes."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    n = 4096
    queries, chains = _headline_inputs(n)
    m = ScanMatcher(None, semantics="yagpy")
    b = m.make_pairs_batch(queries, chains)
    b.run_async(True, True, slot=0)
    per, _, _ = b.wait(0)
    cnt = m.debug_counters()
    assert cnt["last_correlate"] == "correlate_region_kernel" and (cnt["yag_fast_items"], cnt["yag_fallback_items"]) == (n, 0), cnt
    b.run_async(True, True, slot=1)
    again, _, _ = b.wait(1)
    for f in ("response", "pose", "cov", "hypotheses"):
        assert np.array_equal(per.array[f], again.array[f]), f
    ms = ScanMatcher(None, semantics="yagpy")
    ms.debug_option(46, 0)
    for i in range(n):
        s = ms.match_scan(queries[i], chains[i], True, True)
        r = per.array[i]
        assert s.response == r["response"] and np.array_equal(np.array(s.covariance).ravel(), r["cov"]), i
        assert (s.best_pose.x, s.best_pose.y, s.best_pose.euler[-1]) == tuple(r["pose"].tolist()), i
        assert s.meta["hypotheses"] == r["hypotheses"], i
    o = orc.Oracle(None, "yagpy")
    for i in sorted(np.random.default_rng(2048).choice(n, size=16, replace=False).tolist()):
        ro = o.match_scan(_plain(queries[i]), [_plain(s) for s in chains[i]], True, True)
        r = per[i]
        assert abs(r.response - ro["response"]) <= 1e-12, (i, r.response, ro["response"])
        bp = r.best_pose
        np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], ro["pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
        assert np.array_equal(m.debug_sums(0, item=i, dims=r.meta["coarse_dims"]), o.sums(0)), i
        assert np.array_equal(m.debug_sums(1, item=i, dims=r.meta["fine_dims"]), o.sums(1)), i


def test_match_pairs_argument_errors():
    from yag_slam_amd import _capi
    from yag_slam_amd.scan_matching import ScanMatcher
    queries, chains = _pairs_workload(2)
    m = ScanMatcher()
    with pytest.raises(ValueError):
        m.match_pairs(queries, chains[:1])
    assert len(m.match_pairs([], [])) == 0
    import ctypes as C
    hq = (C.c_void_p * 2)(queries[0].native(0), None)
    hs = (C.c_void_p * 1)(chains[0][0].native(0))
    co = (C.c_int32 * 3)(0, 1, 1)
    per = (_capi.YmResult * 2)()
    assert m._lib.ym_match_pairs(m._m, hq, hs, co, 2, 1, 1, per) == -1  # YM_ERR_INVALID
    assert "query 1 is null" in _capi.last_error()


def test_match_pairs_in_the_python_semantics_and_on_the_loop_lattice():
    """ym_match_pairs outside the default Karto lattice: the reference's in-tree Python semantics (yagpy device path) and the
    loop-closure config (41 x 41 x 21: the gather correlate, one list set per query) -- every item is its single call."""
    from yag_slam_amd.scan_matching import ScanMatcher
    queries, chains = _pairs_workload(72)
    for kw, n, pen, fine in ((dict(semantics="yagpy"), 9, True, True), (dict(loop=True), 72, False, False), (dict(loop=True), 9, False, False)):
        m, ms = ScanMatcher(None, **kw), ScanMatcher(None, **kw)
        per = m.match_pairs(queries[:n], chains[:n], pen, fine)
        for i in range(n):
            s = ms.match_scan(queries[i], chains[i], pen, fine)
            a = per[i]
            assert a.response == s.response and a.covariance == s.covariance and a.meta == s.meta, (kw, i)
            assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (s.best_pose.x, s.best_pose.y, s.best_pose.euler[-1]), (kw, i)
