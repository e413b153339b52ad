"""HIP path vs CPU oracle ("karto" semantics) through the C ABI, on a real MI355X.
Integer work (grid bytes, correlation sums) must be bit-exact; fp64 results within 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.util import PlainScan, cfg2_scans  # noqa: E402


def _mk_native(plain):
    from yag_slam_amd.models import LocalizedRangeScan
    p = plain.corrected_pose
    return LocalizedRangeScan(plain.ranges, plain.min_angle, plain.max_angle, plain.angle_increment,
                              plain.min_range, plain.max_range, plain.range_threshold, p.x, p.y, p.euler[-1])


def _has_experimental_forms():
    """The correlate forms that lost to correlate_region_kernel (debug option 32 = 2, 3, 4) are compiled only into builds made
    with -DYM_EXPERIMENTAL; a default build refuses the option with YM_ERR_UNSUPPORTED and the tests leave those forms out."""
    from yag_slam_amd import _capi
    from yag_slam_amd.scan_matching import ScanMatcher
    m = ScanMatcher()
    try:
        m.debug_option(32, 2)
        return True
    except _capi.YmError as e:
        assert e.code == -4
        return False
    finally:
        m.close()


def compare(cfg, query, base, penalty, fine, loop=False, resident=True, check_grid=True):
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    o = orc.Oracle(cfg, "karto", loop=loop)
    ro = o.match_scan(query, base, penalty, fine)
    m = ScanMatcher(cfg, loop=loop)
    m.debug_option(12, 1)  # (keep the integer sums whatever the lattice size)
    if resident:
        rq, rb = _mk_native(query), [_mk_native(b) for b in base]
    else:
        rq, rb = query, base
    r = m.match_scan(rq, rb, penalty, fine)
    assert r.meta["n_query_points"] == ro["n_query_points"]
    assert r.meta["expansions"] == ro["expansions"]
    assert r.meta["hypotheses"] == ro["hypotheses"]
    if ro["n_query_points"] > 0:
        assert r.meta["coarse_dims"] == ro["coarse_dims"]
        if ro["expansions"] == 0:
            # intermediates of the last pipeline run == the oracle's only run
            ql = m.debug_query_local()
            np.testing.assert_allclose(ql, o.query_local(), rtol=0, atol=1e-11)
            if check_grid:
                g, info = m.debug_grid()
                og, oinfo = o.grid_u8()
                assert info.storage_w == oinfo["width"] and info.roi_x == oinfo["roi"][0] and info.roi_w == oinfo["roi"][2]
                sub = og[info.origin_y:info.origin_y + info.height, info.origin_x:info.origin_x + info.width]
                assert np.array_equal(g, sub), "grid window differs: %d cells" % int((g != sub).sum())
            s0 = m.debug_sums(0, dims=r.meta["coarse_dims"])
            assert np.array_equal(s0, o.sums(0)), "coarse sums differ"
            if fine:
                s1 = m.debug_sums(1, dims=r.meta["fine_dims"])
                assert np.array_equal(s1, o.sums(1)), "fine sums differ"
    assert abs(r.response - ro["response"]) <= 1e-12
    bp = r.best_pose
    np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], ro["pose"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.array(r.covariance), ro["cov"], rtol=1e-9, atol=1e-15)
    return r, ro


@pytest.mark.parametrize("penalty,fine", [(True, True), (False, True), (True, False), (False, False)])
def test_cfg2_default(penalty, fine):
    q, base = cfg2_scans()
    r, ro = compare(None, q, base, penalty, fine)
    if fine and not penalty:
        assert abs(r.best_pose.x - 3.07) < 0.011 and abs(r.best_pose.y - 3.04) < 0.011


def test_cfg2_descriptor_entry_matches_resident():
    q, base = cfg2_scans()
    compare(None, q, base, True, True, resident=False)


def test_cfg2_dirty_scans():
    q, base = cfg2_scans(dirty=True)
    compare(None, q, base, True, True)


def test_loop_config_coarse_only():
    q, base = cfg2_scans()
    compare(None, q, base, False, False, loop=True)
    compare(None, q, base, False, True, loop=True)


def test_rotated_query_and_mixed_headings():
    from yag_slam_amd import synth
    scene = synth.Scene()
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 20.0, p)
    poses = [(4.0 + 0.08 * i, 2.6 + 0.03 * i, 2.0 + 0.05 * i) for i in range(6)]
    base = [mk(scene.scan_ranges(p, index=200 + i), p) for i, p in enumerate(poses)]
    q = mk(scene.scan_ranges((4.45, 2.8, 2.33), index=210), (4.4, 2.75, 2.28))
    compare(None, q, base, True, True)
    # heading across the +-pi wrap
    poses = [(4.0, 3.0, 3.10 + 0.01 * i) for i in range(3)]
    base = [mk(scene.scan_ranges(p, index=220 + i), p) for i, p in enumerate(poses)]
    q = mk(scene.scan_ranges((4.02, 3.01, -3.12), index=230), (4.0, 3.0, 3.14))
    compare(None, q, base, True, True)


def test_small_lattice_config():
    from tests.util import load_case
    c = load_case("small_pen1_fine1")
    cfg = dict(c["cfg"], search_size=0.32)  # Karto needs search_size/resolution even
    compare(cfg, c["query"], c["base"], True, True)
    c = load_case("small_dirty_rot")
    compare(cfg, c["query"], c["base"], False, True)


def test_invalid_configs_fail_like_the_reference():
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd._capi import YmError
    with pytest.raises(AssertionError):  # helpers.py:370
        ScanMatcher(dict(resolution=0.01, smear_deviation=0.2))
    with pytest.raises(YmError):  # odd search_size/resolution: Karto throws in the probability search
        ScanMatcher(dict(search_size=0.3, resolution=0.02, smear_deviation=0.04))


def test_reference_smoke_input():
    # /root/reference/test.py:29-38: 230 beams x 3.0 m, base (0,0,0), query (1.0,0,1.57), penalize + refine
    sensor = dict(min_angle=-1.0, angle_increment=float(np.deg2rad(0.5)), min_range=0.0, range_threshold=5.0)
    mk = lambda p: PlainScan([3.0] * 230, sensor["min_angle"], sensor["angle_increment"], 0.0, 5.0, p)
    compare(None, mk((1.0, 0.0, 1.57)), [mk((0.0, 0.0, 0.0))], True, True)


def test_response_expansion_path():
    # a query that sees nothing of the base scan: zero response -> three +20 degree retries
    sensor_mk = lambda r, p: PlainScan(r, -0.5, 0.01, 0.05, 20.0, p)
    base = [sensor_mk(np.full(101, 2.0), (0.0, 0.0, 0.0))]
    q = sensor_mk(np.full(101, 2.0), (10.0, 10.0, 0.0))
    r, ro = compare(dict(search_size=0.3, range_threshold=12.0), q, base, True, True)
    assert ro["expansions"] == 3 and r.response == 0.0
    r, ro = compare(dict(search_size=0.3, range_threshold=12.0, use_response_expansion=False), q, base, True, True)
    assert ro["expansions"] == 0


def test_empty_and_ragged_inputs():
    q, base = cfg2_scans()
    # no base scans at all
    compare(None, q, [], True, True)
    # query without a single valid reading: Karto's early return
    qe = PlainScan(np.full(1081, np.nan), q.min_angle, q.angle_increment, q.min_range, 20.0, (3.0, 3.0, 0.0))
    r, ro = compare(None, qe, base[:2], True, True)
    assert r.response == 0.0 and r.covariance[0][0] == 500.0
    # ragged chain: scans of different lengths, one empty, one all over-range
    rag = [base[0],
           PlainScan(base[1].ranges[:300], q.min_angle, q.angle_increment, q.min_range, 20.0, (2.1, 3.0, 0.0)),
           PlainScan(np.zeros(0), q.min_angle, q.angle_increment, q.min_range, 20.0, (2.2, 3.0, 0.0)),
           PlainScan(np.full(500, 25.0), q.min_angle, q.angle_increment, q.min_range, 20.0, (2.3, 3.0, 0.0))]
    compare(None, q, rag, True, True)


def test_batch_matches_single_calls():
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd import synth
    scene = synth.Scene()
    q, base = cfg2_scans()
    chains_p = synth.chain_poses(5, chain_len=4, scene=scene)
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 20.0, p)
    chains = [[_mk_native(mk(scene.scan_ranges(p, index=300 + 10 * c + i), p)) for i, p in enumerate(ch)]
              for c, ch in enumerate(chains_p)]
    m = ScanMatcher(None, loop=True)
    nq = _mk_native(q)
    per, best = m.match_scan_batch(nq, chains, False, False)
    singles = [m.match_scan(nq, ch, False, False) for ch in chains]
    for a, b in zip(per, singles):
        assert a.response == b.response
        assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
        assert a.covariance == b.covariance
    assert best == int(np.argmax([s.response for s in singles]))


@pytest.mark.parametrize("finish_form", [0, 1])
def test_large_batch_matches_single_calls(finish_form):
    """Batches of 8 or more items take different kernels than a single match (256-thread prepare blocks, the raster's
    tile work list, the one-block finish kernel; option 6 = 1 forces the per-angle fine/final pair): every result
    must still be the single call's, bit for bit -- penalty, refinement, ragged and empty chains included."""
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd import synth
    scene = synth.Scene()
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    chains_p = synth.chain_poses(6, chain_len=5, scene=scene)
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 20.0, p)
    far = [[_mk_native(mk(scene.scan_ranges(p, index=500 + 10 * c + i), p)) for i, p in enumerate(ch)]
           for c, ch in enumerate(chains_p)]
    chains = [nb, nb[:3], nb[2:9], far[1], [], nb[::-1], far[2], nb[4:5], far[3], nb[1:], far[4]]
    m = ScanMatcher()
    m.debug_option(6, finish_form)
    for pen, fine in ((True, True), (False, False)):
        per, best = m.match_scan_batch(nq, chains, pen, fine)
        m.debug_option(6, 0)
        singles = [m.match_scan(nq, ch, pen, fine) for ch in chains]
        m.debug_option(6, finish_form)
        for a, b in zip(per, singles):
            assert a.response == b.response
            assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
            assert a.covariance == b.covariance
            assert a.meta["hypotheses"] == b.meta["hypotheses"] and a.meta["expansions"] == b.meta["expansions"]
        assert best == int(np.argmax([s.response for s in singles]))
    # a second, different call on the same matcher: the tile work list must clear what the first one left behind
    g1 = m.match_scan_batch(nq, [far[5]] * 9, True, True)[0][0]
    g2 = m.match_scan(nq, far[5], True, True)
    assert g1.response == g2.response and g1.covariance == g2.covariance


def test_finish_kernel_single_item_matches_pair():
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    m = ScanMatcher()
    ref = m.match_scan(nq, nb, True, True)
    fs_ref = m.debug_sums(1, dims=ref.meta["fine_dims"])
    m.debug_option(6, 2)
    for threads in (1024, 256):  # both block sizes of the one-block finish kernel: the same bits
        m.debug_option(11, threads)
        got = m.match_scan(nq, nb, True, True)
        assert got.response == ref.response and got.covariance == ref.covariance
        assert (got.best_pose.x, got.best_pose.y, got.best_pose.euler[-1]) == (ref.best_pose.x, ref.best_pose.y, ref.best_pose.euler[-1])
        assert np.array_equal(m.debug_sums(1, dims=got.meta["fine_dims"]), fs_ref)


def test_async_pipeline_matches_sync():
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    m = ScanMatcher()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    ref = m.match_scan(nq, nb, True, True)
    for s in range(8):
        m.match_scan_async(nq, nb, True, True, slot=s)
    for s in range(8):
        r = m.wait(s)
        assert r.response == ref.response and r.covariance == ref.covariance


# ------------------------------------------------------------------------------------------------
# "yagpy" semantics on the device against the golden vectors produced by the REFERENCE's own
# Python matcher (tests/golden/make_golden.py): same inputs, the reference's outputs.
GOLDEN_CASES = ["small_pen0_fine1", "small_pen1_fine1", "small_pen1_fine0", "small_dirty_rot", "testpy_flat",
                "cfg2_pen1_fine1", "cfg2_pen0_fine1", "cfg2_pen1_fine0"]


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_yagpy_device_matches_reference_goldens(name):
    from tests.util import load_case
    from yag_slam_amd.scan_matching import ScanMatcher
    c = load_case(name)
    z = c["z"]
    m = ScanMatcher(c["cfg"], semantics="yagpy")
    r = m.match_scan(c["query"], c["base"], c["penalty"], c["do_fine"])
    # correlation grid: the device window against the same rectangle of the reference's grid,
    # in the integer form the reference scores with, int(100 * cell)  (helpers.py:142-145)
    G = int(z["grid_size"])
    g, info = m.debug_grid()
    assert info.storage_w == G
    full = np.zeros((G, G), dtype=np.uint8)
    full[z["grid_nz_y"], z["grid_nz_x"]] = (100 * z["grid_nz_val"]).astype(np.int64)
    sub = full[info.origin_y:info.origin_y + info.height, info.origin_x:info.origin_x + info.width]
    assert np.array_equal(g, sub), "grid window differs in %d cells" % int((g != sub).sum())
    if "coarse_sums" in z.files:  # integer work: bit-exact against the reference's own scoring function
        s0 = m.debug_sums(0, dims=r.meta["coarse_dims"])
        assert s0.shape == z["coarse_sums"].shape
        assert np.array_equal(s0.astype(np.int64), z["coarse_sums"])
    assert abs(r.response - float(z["response"])) <= 1e-12
    bp = r.best_pose
    np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], z["best_pose"], rtol=0, atol=1e-9)
    cov, got = z["covariance"], np.array(r.covariance)
    if np.all(np.isfinite(cov)):
        np.testing.assert_allclose(got, cov, rtol=1e-9, atol=1e-15)
    else:  # zero response: the reference divides by zero; same NaN/inf pattern expected
        assert np.array_equal(np.isfinite(got), np.isfinite(cov)) and np.array_equal(np.isnan(got), np.isnan(cov))


# Round 6: the HOT correlate kernels against reference-made numbers.  In "yagpy" semantics the coarse pass's integer sums come
# from the production correlate kernels wherever the item's roundings provably form a lattice (ym_k_yagpy.hpp, yag_lattice_kernel):
# correlate_kernel on single matches, correlate_region_kernel on batches over lattices up to 26 x 32, gather_kernel on the others --
# the same selection as in Karto semantics.  Every route must give the sum volume the reference's own score_world_points_on_grid
# produced (tests/golden/make_golden.py: coarse_sums of the small cases; make_golden_sums.py: full-size volumes of both passes),
# with no item falling back to the pair-by-pair kernel.
SUM_CASES = ["sums_cfg2", "sums_dirty_rot", "sums_far", "sums_loop", "sums_near_threshold"]
SUM_SMALL = ["small_pen0_fine1", "small_pen1_fine1", "small_dirty_rot"]
YAG_ROUTES = ["pairwise", "direct", "region", "gather"]


@pytest.mark.parametrize("route", YAG_ROUTES)
@pytest.mark.parametrize("name", SUM_CASES + SUM_SMALL)
def test_yagpy_coarse_sums_through_the_production_kernels(name, route):
    from tests.util import load_case
    from yag_slam_amd.scan_matching import ScanMatcher
    c = load_case(name)
    z = c["z"]
    want = z["coarse_sums"].astype(np.int64)
    nt, ny, nx = want.shape
    m = ScanMatcher(c["cfg"], semantics="yagpy")
    q, base = _mk_native(c["query"]), [_mk_native(b) for b in c["base"]]

    def check_result(r):
        assert r.meta["coarse_dims"] == (nx, ny, nt)
        assert abs(r.response - float(z["response"])) <= 1e-12
        bp = r.best_pose
        np.testing.assert_allclose([bp.x, bp.y, bp.euler[-1]], z["best_pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(r.covariance), z["covariance"], rtol=1e-9, atol=1e-15)

    if route in ("pairwise", "direct"):
        if route == "pairwise":
            m.debug_option(46, 0)  # the Python rule as written: every (hypothesis, point) pair rounded on its own
        elif name in SUM_CASES[::2]:
            m.debug_option(46, 2)  # the fine pass's rows byte by byte (what a row takes whose columns do not fit one 8-byte read)
        r = m.match_scan(q, base, c["penalty"], c["do_fine"])
        cnt = m.debug_counters()
        assert cnt["last_correlate"] == (None if route == "pairwise" else "correlate_kernel")
        assert (cnt["yag_fast_items"], cnt["yag_fallback_items"]) == ((0, 0) if route == "pairwise" else (1, 0))
        assert np.array_equal(m.debug_sums(0, dims=r.meta["coarse_dims"]).astype(np.int64), want)
        if c["do_fine"] and "fine_sums" in z.files:
            assert np.array_equal(m.debug_sums(1, dims=r.meta["fine_dims"]).astype(np.int64), z["fine_sums"])
        check_result(r)
        return
    # batches: nine items -- seven copies of the golden problem, one with a shorter chain, one copy more -- through the LDS correlates
    m.debug_option(28, 8)  # (both LDS correlates from 8 items on; production takes them from 48 / 64)
    if route == "gather":
        m.debug_option(14, 4)  # the gather correlate also where the region correlate would run
    chains = [base] * 7 + [base[:2]] + [base]
    per, _ = m.match_scan_batch(q, chains, c["penalty"], c["do_fine"])
    cnt = m.debug_counters()
    fits_region = name != "sums_loop"  # (its launch lattice is 41 x 41: beyond the region correlate's 26 x 32, the gather correlate's domain)
    assert cnt["last_correlate"] == ("correlate_region_kernel" if route == "region" and fits_region else "gather_kernel")
    assert (cnt["yag_fast_items"], cnt["yag_fallback_items"]) == (9, 0)
    for i in (0, 3, 6, 8):
        assert np.array_equal(m.debug_sums(0, item=i, dims=per[i].meta["coarse_dims"]).astype(np.int64), want), i
        if c["do_fine"] and "fine_sums" in z.files:  # (yag_fine_kernel inside a batch, against the reference's own fine volume)
            assert np.array_equal(m.debug_sums(1, item=i, dims=per[i].meta["fine_dims"]).astype(np.int64), z["fine_sums"]), i
        check_result(per[i])
    # the shorter chain: against the pair-by-pair kernel (a single call with the production routes off)
    ref = ScanMatcher(c["cfg"], semantics="yagpy")
    ref.debug_option(46, 0)
    r7 = ref.match_scan(q, base[:2], c["penalty"], c["do_fine"])
    assert np.array_equal(m.debug_sums(0, item=7, dims=per[7].meta["coarse_dims"]), ref.debug_sums(0, dims=r7.meta["coarse_dims"]))
    assert per[7].response == r7.response and per[7].covariance == r7.covariance
    assert (per[7].best_pose.x, per[7].best_pose.y, per[7].best_pose.euler[-1]) == (r7.best_pose.x, r7.best_pose.y, r7.best_pose.euler[-1])


def test_yagpy_irregular_items_fall_back_to_the_pairwise_kernel():
    """A reading placed so that hypothesis (0, 0) rounds a TIE (x.5 cells) cannot be proven regular by the guard; the exhaustive
    check then decides.  Either way the results must be the pair-by-pair kernel's, and the counters must say what happened."""
    from tests.util import load_case
    from yag_slam_amd.scan_matching import ScanMatcher
    c = load_case("small_pen1_fine1")
    cfg = dict(c["cfg"], resolution=0.25, smear_deviation=0.25, search_size=1.0, range_threshold=4.0)
    # poses and readings that are exact binary fractions: sums like 0.125 + k * 0.25 sit exactly on cell boundaries
    mk = lambda r, p: _mk_native(PlainScan(r, 0.0, np.pi / 6, 0.05, 4.0, p))
    base = [mk(np.full(7, 1.125), (0.0, 0.0, 0.0))]
    q = mk(np.array([1.125, 1.0, 0.875, 1.125, 1.25, 1.125, 1.0]), (0.125, 0.0, 0.0))
    fast, slow = ScanMatcher(cfg, semantics="yagpy"), ScanMatcher(cfg, semantics="yagpy")
    slow.debug_option(46, 0)
    a, b = fast.match_scan(q, base, True, True), slow.match_scan(q, base, True, True)
    cnt = fast.debug_counters()
    assert cnt["yag_pairs_checked"] > 0, cnt  # ties were met ...
    assert cnt["yag_fast_items"] + cnt["yag_fallback_items"] == 1
    assert a.meta["coarse_dims"] == b.meta["coarse_dims"]
    assert np.array_equal(fast.debug_sums(0, dims=a.meta["coarse_dims"]), slow.debug_sums(0, dims=b.meta["coarse_dims"]))  # ... and nothing changed
    assert a.meta["fine_dims"] == b.meta["fine_dims"]  # the fine pass by rows meets the same ties (its step is one cell of 0.25)
    assert np.array_equal(fast.debug_sums(1, dims=a.meta["fine_dims"]), slow.debug_sums(1, dims=b.meta["fine_dims"]))
    assert a.response == b.response and a.covariance == b.covariance


def test_yagpy_batch_with_regular_and_tie_items_side_by_side():
    """One enqueue through the LDS correlates that holds items the lattice proof accepts at once, items with exact rounding ties that only
    the exhaustive check can decide, and -- whatever that check says -- the same results as the pair-by-pair kernel for every item."""
    from tests.util import load_case
    from yag_slam_amd.scan_matching import ScanMatcher
    c = load_case("small_pen1_fine1")
    cfg = dict(c["cfg"], resolution=0.25, smear_deviation=0.25, search_size=1.0, range_threshold=4.0)
    mk = lambda r, p: _mk_native(PlainScan(r, 0.0, np.pi / 6, 0.05, 4.0, p))
    base = [mk(np.full(7, 1.125), (0.0, 0.0, 0.0)), mk(np.full(7, 1.3), (0.1, 0.05, 0.02))]
    ties = mk(np.array([1.125, 1.0, 0.875, 1.125, 1.25, 1.125, 1.0]), (0.125, 0.0, 0.0))   # binary fractions: sums on cell boundaries
    plain = mk(np.array([1.13, 1.02, 0.91, 1.17, 1.21, 1.09, 1.04]), (0.113, 0.021, 0.013))
    queries = [ties, plain] * 5
    chains = [base, base[:1]] * 5
    fast, slow = ScanMatcher(cfg, semantics="yagpy"), ScanMatcher(cfg, semantics="yagpy")
    fast.debug_option(28, 8)
    slow.debug_option(46, 0)
    a, b = fast.match_pairs(queries, chains, True, True), slow.match_pairs(queries, chains, True, True)
    cnt = fast.debug_counters()
    assert cnt["last_correlate"] in ("correlate_region_kernel", "gather_kernel")
    assert cnt["yag_fast_items"] + cnt["yag_fallback_items"] == 10 and cnt["yag_fast_items"] >= 5 and cnt["yag_pairs_checked"] > 0, cnt
    for i in range(10):
        assert a[i].meta["coarse_dims"] == b[i].meta["coarse_dims"]
        assert np.array_equal(fast.debug_sums(0, item=i, dims=a[i].meta["coarse_dims"]), slow.debug_sums(0, item=i, dims=b[i].meta["coarse_dims"])), i
        assert a[i].meta["fine_dims"] == b[i].meta["fine_dims"]
        assert np.array_equal(fast.debug_sums(1, item=i, dims=a[i].meta["fine_dims"]), slow.debug_sums(1, item=i, dims=b[i].meta["fine_dims"])), i
        assert (a[i].response, a[i].covariance) == (b[i].response, b[i].covariance), i


@pytest.mark.parametrize("seed", [1, 3])
def test_yagpy_production_routes_on_random_problems(seed):
    """tests/yag_soak.py: random problems (poses along a trajectory, a third of the seeds kilometres from the origin, dirty readings, three
    configurations, single matches and batches through every correlate) solved with the coarse sums from the production kernels and pair
    by pair: identical results and sum volumes, and the proof accepted every item (scripts/dev/r06_yag_soak.py runs many seeds)."""
    from tests import yag_soak
    bad, totals = yag_soak.run(seed, 24, verbose=False)
    assert bad == 0
    assert totals["yag_fast_items"] > 0 and totals["yag_fallback_items"] == 0, totals


def test_yagpy_device_matches_oracle_both_passes():
    from oracle import oracle as orc
    from tests.util import load_case
    from yag_slam_amd.scan_matching import ScanMatcher
    c = load_case("small_dirty_rot")
    o = orc.Oracle(c["cfg"], "yagpy")
    ro = o.match_scan(c["query"], c["base"], True, True)
    m = ScanMatcher(c["cfg"], semantics="yagpy")
    r = m.match_scan(c["query"], c["base"], True, True)
    assert r.meta["coarse_dims"] == ro["coarse_dims"] and r.meta["fine_dims"] == ro["fine_dims"]
    assert r.meta["hypotheses"] == ro["hypotheses"]
    assert np.array_equal(m.debug_sums(0, dims=r.meta["coarse_dims"]), o.sums(0))
    assert np.array_equal(m.debug_sums(1, dims=r.meta["fine_dims"]), o.sums(1))
    assert abs(r.response - ro["response"]) <= 1e-12


def test_order_dependent_smear_boundary():
    # smear_deviation = 10 * resolution: the four neighbours of an occupied cell are 100 as well, so
    # Karto's "value already set" skip makes the raster depend on point order
    from tests.util import load_case
    c = load_case("small_pen1_fine1")
    cfg = dict(c["cfg"], search_size=0.32, resolution=0.005, smear_deviation=0.05)
    from oracle import oracle as orc
    assert (orc.kernel_karto(0.005, 0.05) == 100).sum() == 5
    compare(cfg, c["query"], c["base"], True, True)
    q, base = cfg2_scans(range_threshold=12.0)
    compare(dict(resolution=0.005, smear_deviation=0.05, range_threshold=12.0), q, base[:4], True, True)


def test_order_dependent_smear_in_a_batch():
    # the select kernel, the tile work list and the one-block finish kernel together: every item of a batch of 9
    # equals its single call when smear_deviation = 10 * resolution
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans(range_threshold=12.0)
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    m = ScanMatcher(dict(resolution=0.005, smear_deviation=0.05, range_threshold=12.0))
    chains = [nb[:4], nb[2:6], nb[:2], nb[5:], nb[1:4], nb[:4][::-1], nb[3:4], nb[4:9], nb[:6]]
    # a full batch first: it leaves real cells in every chain slot, which the ragged batch's unused slots must not
    # feed to the select kernel
    m.match_scan_batch(nq, [nb[4:]] * 9, True, True)
    per, best = m.match_scan_batch(nq, chains, True, True)
    for ch, a in zip(chains, per):
        b = m.match_scan(nq, ch, True, True)
        assert a.response == b.response and a.covariance == b.covariance
        assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])


def test_stress_config_full_lattice():
    # BASELINE configs[4]: search_size=2.0 resolution=0.005 coarse_angle_offset=0.785 (201x201x46 + 99
    # hypotheses); every one of the 1.86 M integer sums is compared with the oracle
    q, base = cfg2_scans()
    cfg = dict(search_size=2.0, resolution=0.005, coarse_search_angle_offset=0.785)
    r, ro = compare(cfg, q, base, True, True, check_grid=True)
    assert r.meta["coarse_dims"] == (201, 201, 46) and r.meta["hypotheses"] == 1858545


def test_direct_correlate_sums_are_the_oracles():
    # the direct (global-load) coarse correlate of a single match: the same integer sums as the oracle; and a batch of
    # a few chains (longer beam chunks per wave) against single calls
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    o = orc.Oracle(None, "karto")
    ro = o.match_scan(q, base, True, True)
    m = ScanMatcher()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    r = m.match_scan(nq, nb, True, True)
    assert np.array_equal(m.debug_sums(0, dims=r.meta["coarse_dims"]), o.sums(0))
    assert abs(r.response - ro["response"]) <= 1e-12
    per, best = m.match_scan_batch(nq, [nb[:5], nb[3:], nb], True, True)
    m2 = ScanMatcher()
    for ch, p in zip([nb[:5], nb[3:], nb], per):
        s = m2.match_scan(nq, ch, True, True)
        assert s.response == p.response and s.covariance == p.covariance


# ------------------------------------------------------------------------------------------------
# size-independent properties at full size
def test_determinism_and_translation_invariance():
    """Same inputs -> bitwise the same outputs on every run; moving every pose by a whole number of grid cells
    moves the answer by exactly that much and leaves responses/covariances unchanged (the grid is anchored at
    the query pose, so the rasterised window and every integer sum are identical)."""
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    m = ScanMatcher()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    r1 = m.match_scan(nq, nb, True, True)
    s1 = m.debug_sums(0, dims=r1.meta["coarse_dims"])
    for _ in range(5):
        r = m.match_scan(nq, nb, True, True)
        assert r.response == r1.response and r.covariance == r1.covariance
        assert (r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1]) == (r1.best_pose.x, r1.best_pose.y, r1.best_pose.euler[-1])
    # shift by (+64, -32) cells of 0.01 m = binary-exact offsets keep every fp64 difference identical
    dx, dy = 0.5, -0.25
    mk = lambda s: PlainScan(s.ranges, s.min_angle, s.angle_increment, s.min_range, s.range_threshold,
                             (s.corrected_pose.x + dx, s.corrected_pose.y + dy, s.corrected_pose.euler[-1]))
    r2 = m.match_scan(mk(q), [mk(b) for b in base], True, True)
    s2 = m.debug_sums(0, dims=r2.meta["coarse_dims"])
    assert np.array_equal(s1, s2)
    assert abs(r2.response - r1.response) <= 1e-15
    assert abs(r2.best_pose.x - r1.best_pose.x - dx) <= 1e-12 and abs(r2.best_pose.y - r1.best_pose.y - dy) <= 1e-12
    np.testing.assert_allclose(np.array(r2.covariance), np.array(r1.covariance), rtol=1e-9, atol=1e-18)


def test_maximum_scan_size_and_limits():
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd._capi import YmError
    from yag_slam_amd import synth
    scene = synth.Scene()
    n = 6000  # YM_MAX_BEAMS
    inc = 2 * np.pi / n
    mk = lambda p, idx, nb: PlainScan(scene.scan_ranges(p, index=idx, n_beams=nb, min_angle=-np.pi, inc=inc), -np.pi, inc, 0.05, 20.0, p)
    cfg = dict(search_size=0.32, resolution=0.02, smear_deviation=0.04, range_threshold=12.0)
    base = [mk((3.0 + 0.1 * i, 3.0, 0.1 * i), 400 + i, n) for i in range(3)]
    q = mk((3.12, 3.05, 0.12), 410, n)
    q.corrected_pose.x, q.corrected_pose.y = 3.1, 3.0
    compare(cfg, q, base, True, True)
    with pytest.raises(YmError, match="limit"):
        ScanMatcher(cfg).match_scan(PlainScan(np.full(6001, 2.0), -np.pi, inc, 0.05, 20.0, (3, 3, 0)), base[:1], True, True)


def test_against_karto_wheel_if_present():
    """Optional: wherever the reference's native dependency is installed, compare against it directly
    (north_star tolerances: response 1e-4, pose 1e-3 m / 1e-3 rad) -- on the cfg2 problem and on the reference's own
    smoke input (/root/reference/test.py:29-38: 230 beams of 3.0 m, base (0,0,0), query (1.0,0,1.57), penalize +
    refine).  This is the one test that could pin the Karto branch of the oracle's eighteen switches."""
    ks = pytest.importorskip("karto_scanmatcher")
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.config import default_config
    cfg = ks.ScanMatcherConfig()
    for k, v in default_config.items():
        setattr(cfg, k, v)
    w = ks.Wrapper(cfg)
    mk = lambda s: ks.LocalizedRangeScan(ks.LaserScanConfig(s.min_angle, s.max_angle, s.angle_increment, s.min_range, s.max_range, s.range_threshold, ""),
                                         list(s.ranges), ks.Pose2(s.corrected_pose.x, s.corrected_pose.y, s.corrected_pose.euler[-1]),
                                         ks.Pose2(s.corrected_pose.x, s.corrected_pose.y, s.corrected_pose.euler[-1]), 0, 0.0)
    q, base = cfg2_scans()
    flat = lambda p: PlainScan([3.0] * 230, -1.0, float(np.deg2rad(0.5)), 0.0, 5.0, p)
    for query, chain in ((q, base), (flat((1.0, 0.0, 1.57)), [flat((0.0, 0.0, 0.0))])):
        ref = w.match_scan(mk(query), [mk(b) for b in chain], True, True)
        r = ScanMatcher().match_scan(query, chain, True, True)
        assert abs(r.response - ref.response) <= 1e-4
        assert abs(r.best_pose.x - ref.best_pose.x) <= 1e-3 and abs(r.best_pose.y - ref.best_pose.y) <= 1e-3
        assert abs(r.best_pose.euler[-1] - ref.best_pose.yaw) <= 1e-3
        for i in range(3):
            assert abs(r.covariance[i][i] - ref.covariance[i][i]) <= 1e-4 * max(1.0, abs(ref.covariance[i][i]))


def test_c_abi_error_behaviour():
    """The boundary never throws: bad calls return a negative YM_ERR_* and leave a message (include/yagmatch.h)."""
    import ctypes as C
    from yag_slam_amd import _capi
    from yag_slam_amd.scan_matching import ScanMatcher
    L = _capi.lib()
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    m = ScanMatcher()
    res = _capi.YmResult()
    # collecting a slot nothing was submitted to
    assert L.ym_wait(m._m, 3, C.byref(res)) == -6 and b"slot" in L.ym_last_error()
    assert L.ym_wait(m._m, 10 ** 6, C.byref(res)) == -1
    # a slot cannot be reused before it is collected
    m.match_scan_async(nq, nb, True, True, slot=1)
    arr = (C.c_void_p * len(nb))(*[s.native(0) for s in nb])
    assert L.ym_match_scans_async(m._m, nq.native(0), arr, len(nb), 1, 1, 1) == -6
    r1 = m.wait(1)
    assert r1.response == m.match_scan(nq, nb, True, True).response
    # null arguments
    assert L.ym_match_scans(m._m, None, arr, len(nb), 1, 1, C.byref(res)) == -1
    assert L.ym_match_scans(m._m, nq.native(0), arr, len(nb), 1, 1, None) == -1
    assert L.ym_match_scans(None, nq.native(0), arr, len(nb), 1, 1, C.byref(res)) == -1
    assert L.ym_debug_option(m._m, 99, 0) == -1
    # no base scans: Karto correlates against an empty grid -> response 0 after the expansion retries, no error
    assert L.ym_match_scans(m._m, nq.native(0), arr, 0, 1, 1, C.byref(res)) == 0
    assert res.response == 0.0 and res.expansions == 3 and res.status == 0
    # an empty batch is a valid request with nothing in it
    co = (C.c_int32 * 1)(0)
    bi = C.c_int32(7)
    best = _capi.YmResult()
    rc = L.ym_match_batch(m._m, nq.native(0), arr, co, 0, 1, 1, None, C.byref(best), C.byref(bi))
    assert rc in (0, -1)
    if rc == 0:
        assert bi.value == -1
    # config and scan accessors round-trip
    cfg = _capi.YmConfig()
    assert L.ym_get_config(m._m, C.byref(cfg)) == 0 and cfg.search_size == 0.5 and cfg.resolution == 0.01
    pose = (C.c_double * 3)()
    h = nb[0].native(0)
    assert L.ym_scan_set_pose(h, 1.25, -2.5, 0.75) == 0 and L.ym_scan_get_pose(h, pose) == 0
    assert tuple(pose) == (1.25, -2.5, 0.75) and L.ym_scan_size(h) == 1081
    p = nb[0].corrected_pose
    assert L.ym_scan_set_pose(h, p.x, p.y, p.euler[-1]) == 0
    # the introspection calls refuse an item the last call did not have
    info = _capi.YmGridInfo()
    assert L.ym_debug_grid_info(m._m, 5, C.byref(info)) == -1


def test_bulk_pose_write_equals_the_per_scan_writes():
    """ym_scans_set_poses / models.set_corrected_poses (graph_slam.py:263-272: every vertex re-posed after an optimisation):
    the same poses on the device twins and the same match as one ym_scan_set_pose per scan; a resident batch enqueued again
    sees the move (its plan is not replayed); a null entry fails the call before anything is written."""
    import ctypes as C
    from yag_slam_amd import _capi, synth
    from yag_slam_amd.models import set_corrected_poses
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    L = _capi.lib()
    m = ScanMatcher()
    q, base = synth.single_match_scans()
    q2, base2 = synth.single_match_scans()
    for s in (q, q2, *base, *base2):
        s.native(0)
    rng = np.random.default_rng(5)
    moved = np.array([(s.corrected_pose.x + rng.normal(0, 0.02), s.corrected_pose.y + rng.normal(0, 0.02),
                       s.corrected_pose.euler[-1] + rng.normal(0, 0.01)) for s in base])
    for s, p in zip(base, moved):                       # one write per scan
        s.corrected_pose = Transform(p[0], p[1], 0.0, p[2])
    batch = m.make_batch(q2, [base2, base2[:5]])
    batch.run_async(True, True)
    before = batch.wait()[0]
    set_corrected_poses(base2, moved)                   # one call
    pose = (C.c_double * 3)()
    for s, p in zip(base2, moved):
        assert L.ym_scan_get_pose(s._native, pose) == 0 and tuple(pose) == tuple(p)
        cp = s.corrected_pose
        assert (cp.x, cp.y, cp.euler[-1]) == tuple(p)
    a, b = m.match_scan(q, base, True, True), m.match_scan(q2, base2, True, True)
    assert a.response == b.response and a.covariance == b.covariance
    assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
    batch.run_async(True, True)
    after = batch.wait()[0]
    assert after[0].response == a.response and before[0].response != after[0].response
    # Transforms instead of an array; scans without a twin are only written on the Python side
    plain = synth.single_match_scans()[1][:3]
    set_corrected_poses(plain + base2[:2], [Transform(1.0 + i, 2.0, 0.0, 0.1) for i in range(5)])
    assert plain[2].corrected_pose.x == 3.0 and plain[0]._native is None
    assert L.ym_scan_get_pose(base2[1]._native, pose) == 0 and tuple(pose) == (5.0, 2.0, 0.1)
    # all or nothing
    hs = (C.c_void_p * 3)(base2[0]._native, None, base2[2]._native)
    xyz = (C.c_double * 9)(*([9.0] * 9))
    assert L.ym_scans_set_poses(hs, xyz, 3) == -1
    assert L.ym_scan_get_pose(base2[0]._native, pose) == 0 and tuple(pose) == (4.0, 2.0, 0.1)
    assert L.ym_scans_set_poses(None, None, 0) == 0


def _random_case(seed):
    """seeded random scene, sensor, poses and matcher config -> (cfg, query, base scans, penalty, fine)"""
    from yag_slam_amd import synth
    rng = np.random.default_rng(9000 + seed)
    scene = synth.Scene(width=float(rng.uniform(6, 10)), height=float(rng.uniform(5, 8)), n_boxes=int(rng.integers(2, 8)),
                        seed=int(rng.integers(1, 10 ** 6)))
    res = float(rng.choice([0.01, 0.02, 0.05]))
    cfg = dict(resolution=res,
               search_size=float(rng.choice([10, 16, 24, 40])) * res,
               smear_deviation=float(rng.choice([1.0, 2.5, 5.0, 8.0])) * res,
               coarse_search_angle_offset=float(rng.choice([0.1745, 0.349, 0.5])),
               coarse_angle_resolution=float(rng.choice([0.0349, 0.0175, 0.05])),
               range_threshold=float(rng.choice([6.0, 12.0, 20.0])),
               use_response_expansion=bool(rng.integers(0, 2)))
    n_beams = int(rng.choice([181, 360, 721, 1081]))
    fov = float(rng.choice([np.pi, 1.5 * np.pi, 2 * np.pi * (1 - 1.0 / n_beams)]))
    min_angle, inc = -fov / 2, fov / (n_beams - 1)
    shift = np.array([float(rng.choice([0.0, -37.3, 1250.0])), float(rng.choice([0.0, 512.25, -88.0]))])
    x0, y0 = rng.uniform(1.5, scene.width - 1.5), rng.uniform(1.5, scene.height - 1.5)
    th0 = float(rng.uniform(-np.pi, np.pi))
    n_base = int(rng.integers(1, 11))
    dirty = bool(rng.integers(0, 2))

    def scan(pose, idx, noise_pose=(0, 0, 0)):
        r = scene.scan_ranges(pose, index=idx, dirty=dirty, n_beams=n_beams, min_angle=min_angle, inc=inc)
        p = (pose[0] + shift[0] + noise_pose[0], pose[1] + shift[1] + noise_pose[1], pose[2] + noise_pose[2])
        return PlainScan(r, min_angle, inc, 0.05, cfg["range_threshold"], p)

    base = []
    for i in range(n_base):
        px = min(max(x0 + 0.08 * i * np.cos(th0), 1.2), scene.width - 1.2)
        py = min(max(y0 + 0.08 * i * np.sin(th0), 1.2), scene.height - 1.2)
        base.append(scan((px, py, th0 + 0.02 * i), 700 + i))
    qx = min(max(x0 + 0.08 * n_base * np.cos(th0), 1.2), scene.width - 1.2)
    qy = min(max(y0 + 0.08 * n_base * np.sin(th0), 1.2), scene.height - 1.2)
    err = (float(rng.normal(0, 0.2 * cfg["search_size"])), float(rng.normal(0, 0.2 * cfg["search_size"])), float(rng.normal(0, 0.05)))
    query = scan((qx, qy, th0 + 0.02 * n_base), 900, noise_pose=err)
    return cfg, query, base, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), rng


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_sweep_against_oracle(seed):
    """Seeded random scenes, sensors, poses and matcher configs: grid bytes and both integer sum volumes bit-exact,
    response / pose / covariance to 1e-12 / 1e-9 -- far-from-origin and negative coordinates, rotated chains, short
    scans, dirty readings, every resolution the reference's configs use."""
    cfg, query, base, pen, fine, _ = _random_case(seed)
    compare(cfg, query, base, pen, fine)


@pytest.mark.parametrize("seed", [100, 101, 102, 103, 104, 105])
def test_random_batches_equal_single_calls(seed):
    """the same random cases as batches of 8..12 chains (sub-chains, reversed, empty): the batch kernels (tile work
    list, 256-thread prepare, one-block finish) give every chain the single call's result, bit for bit"""
    from yag_slam_amd.scan_matching import ScanMatcher
    cfg, query, base, pen, fine, rng = _random_case(seed)
    nq, nb = _mk_native(query), [_mk_native(b) for b in base]
    chains = []
    for _ in range(int(rng.integers(8, 13))):
        kind = int(rng.integers(0, 5))
        lo = int(rng.integers(0, len(nb)))
        hi = int(rng.integers(lo, len(nb))) + 1
        ch = nb[lo:hi]
        chains.append([] if kind == 0 else ch[::-1] if kind == 1 else ch)
    m = ScanMatcher(cfg)
    per, best = m.match_scan_batch(nq, chains, pen, fine)
    singles = [m.match_scan(nq, ch, pen, fine) for ch in chains]
    for a, b in zip(per, singles):
        assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta
        assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
    assert best == int(np.argmax([s.response for s in singles]))


def test_one_matcher_through_changing_batch_sizes():
    """The window memory, its 'tile already zero' flags, the dirty rectangle and the tile work list live across calls.
    One matcher driven through growing / shrinking batches, single calls and moved chains must give what a fresh
    matcher gives for the same call, bit for bit."""
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd import synth
    scene = synth.Scene()
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 20.0, p)
    far = [[_mk_native(mk(scene.scan_ranges(p, index=800 + 10 * c + i), p)) for i, p in enumerate(ch)]
           for c, ch in enumerate(synth.chain_poses(8, chain_len=6, scene=scene))]
    plans = [
        [nb] * 9,
        [far[1], far[2], nb[:4]],
        [far[3]] * 4 + [nb, far[4]] * 4,
        [nb[5:]],
        [far[5], far[6], far[7], nb[:2], nb[2:4], nb[4:6], nb[6:8], nb[8:], far[1]],
        [[]] * 8,
        [nb] * 16,
        [far[2]],
    ]
    m = ScanMatcher()
    for step, chains in enumerate(plans):
        if len(chains) == 1:
            got = [m.match_scan(nq, chains[0], True, True)]
        else:
            got = m.match_scan_batch(nq, chains, True, True)[0]
        fresh = ScanMatcher()
        for ch, a in zip(chains, got):
            b = fresh.match_scan(nq, ch, True, True)
            assert a.response == b.response and a.covariance == b.covariance, (step, len(ch))
            assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
        # and the grid of the last item is exactly a fresh raster's
        g, info = m.debug_grid(len(chains) - 1)
        fg, finfo = fresh.debug_grid(0)
        assert np.array_equal(g, fg), (step, int((g != fg).sum()))
        fresh.close()


def test_distinct_matchers_in_threads():
    """include/yagmatch.h: distinct matchers are independent -- four threads, each with its own matcher and scans,
    matching at the same time, all get the single-threaded answers (ctypes releases the GIL during the calls)."""
    import threading
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    want = {}
    m0 = ScanMatcher()
    nq0, nb0 = _mk_native(q), [_mk_native(b) for b in base]
    for n in (3, 5, 8, 10):
        r = m0.match_scan(nq0, nb0[:n], True, True)
        want[n] = (r.response, r.covariance, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1])
    errors = []

    def work(n):
        try:
            m = ScanMatcher()
            nq, nb = _mk_native(q), [_mk_native(b) for b in base]
            for i in range(40):
                if i % 4 == 3:
                    r = m.match_scan_batch(nq, [nb[:n]] * 9, True, True)[0][4]
                else:
                    r = m.match_scan(nq, nb[:n], True, True)
                got = (r.response, r.covariance, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1])
                if got != want[n]:
                    errors.append((n, i))
        except Exception as e:  # noqa: BLE001
            errors.append((n, repr(e)))

    threads = [threading.Thread(target=work, args=(n,)) for n in (3, 5, 8, 10)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def _sector_scan(scene, pose, index, lo_deg, hi_deg):
    """a scan that only sees the sector [lo, hi] degrees (sensor frame): every other beam is over-range"""
    from yag_slam_amd import synth
    r = scene.scan_ranges(pose, index=index)
    ang = np.degrees(synth.MIN_ANGLE + np.arange(r.shape[0]) * synth.ANGLE_INCREMENT)
    r[(ang < lo_deg) | (ang > hi_deg)] = synth.MAX_RANGE + 1.0
    return _mk_native(PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 20.0, pose))


def test_stale_window_bytes_do_not_survive_a_smaller_batch():
    """Round-1 advisor finding: a call only rasterises (and cleans) items 0..B-1 of the workspace.  Three chains that
    stamp the EAST of the window, then ONE chain, then three chains that stamp the WEST with the same window geometry:
    items 1 and 2 still hold the first call's stamps, outside the rectangle the third call's chains can reach.  Every
    item's grid must equal a fresh matcher's."""
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd import synth
    scene = synth.Scene()
    q, _ = cfg2_scans()
    nq = _mk_native(q)
    east = [[_sector_scan(scene, (3.0 + 0.05 * i, 3.0 + 0.02 * c, 0.0), 900 + 10 * c + i, -20, 20) for i in range(3)] for c in range(3)]
    west = [[_sector_scan(scene, (3.0 + 0.05 * i, 3.0 - 0.02 * c, np.pi), 950 + 10 * c + i, -20, 20) for i in range(3)] for c in range(3)]
    for n_mid in (1, 2):
        m = ScanMatcher()
        m.match_scan_batch(nq, east, True, True)
        m.match_scan_batch(nq, west[:n_mid], True, True) if n_mid > 1 else m.match_scan(nq, west[0], True, True)
        got = m.match_scan_batch(nq, west, True, True)[0]
        for i, ch in enumerate(west):
            fresh = ScanMatcher()
            want = fresh.match_scan(nq, ch, True, True)
            g, _ = m.debug_grid(i)
            fg, _ = fresh.debug_grid(0)
            assert np.array_equal(g, fg), (n_mid, i, int((g != fg).sum()))
            assert got[i].response == want.response and got[i].covariance == want.covariance
            fresh.close()
        m.close()


def test_query_reading_beyond_the_matcher_threshold_is_answered_like_karto():
    """The device window is cut from Karto's grid, which is sized from the MATCHER's range_threshold.  A query whose own
    threshold lets a longer reading through (/root/reference/yag_slam/models.py:110-116: scans carry range_max * 0.9, the
    matcher's default is 20) points outside that grid, where Karto's GetResponse tests the LINEAR index
    (IsUpTo(index, data size)): an offset that leaves the grid sideways wraps into a neighbouring row.  Such a call takes
    the whole of Karto's storage as its window and forms every index with Karto's row pitch: results, grids and sum volumes
    against the oracle (which keeps Karto's full grid) -- single matches, a batch, coarse only and refined, a query far
    enough out that most of its readings leave the grid; and the same readings gated by the scan's own threshold."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    cfg = dict(range_threshold=4.0)
    far = lambda pose: PlainScan(q.ranges, q.min_angle, q.angle_increment, q.min_range, 20.0, pose)
    assert np.nanmax(q.ranges) > 4.0
    for pose in ((3.0, 3.0, 0.0), (3.02, 2.97, 0.4)):
        for pen, fine in ((True, True), (False, False)):
            compare(cfg, far(pose), base[:3], pen, fine)
    # the sums of the coarse volume, and a batch (the direct kernel's per-cell path, score kernel, finish kernel)
    m, o = ScanMatcher(cfg), orc.Oracle(cfg, "karto")
    fq, nb = _mk_native(far((3.0, 3.0, 0.0))), [_mk_native(b) for b in base[:3]]
    r = m.match_scan(fq, nb, True, True)
    ro = o.match_scan(far((3.0, 3.0, 0.0)), base[:3], True, True)
    assert np.array_equal(m.debug_sums(0, dims=r.meta["coarse_dims"]), o.sums(0))
    assert np.array_equal(m.debug_sums(1, dims=r.meta["fine_dims"]), o.sums(1))
    chains = [nb, nb[:2], nb[1:], nb[::-1], nb[:1], nb[2:], nb, nb[1:2]]
    per, best = m.match_scan_batch(fq, chains, True, True)
    for ch, p in zip([base[:3], base[:2], base[1:3], base[:3][::-1], base[:1], base[2:3], base[:3], base[1:2]], per):
        po = o.match_scan(far((3.0, 3.0, 0.0)), ch, True, True)
        assert abs(p.response - po["response"]) <= 1e-12
        np.testing.assert_allclose([p.best_pose.x, p.best_pose.y, p.best_pose.euler[-1]], po["pose"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.array(p.covariance), po["cov"], rtol=1e-9, atol=1e-15)
    # the same readings gated by the scan's own threshold take the ordinary window
    ok_q = PlainScan(q.ranges, q.min_angle, q.angle_increment, q.min_range, 4.0, (3.0, 3.0, 0.0))
    compare(cfg, ok_q, base[:3], True, True)


def test_point_cache_is_invisible():
    """A matcher keeps the world point readings and the trigger chain of resident base scans per (scan, pose), as Karto's
    LocalizedRangeScan keeps m_PointReadings until the pose is set again.  Hits, pose changes, one scan in several
    chains, a scan moved between two uses, the arena growing and (tiny limit) starting over: every call must equal a
    matcher that never caches, bit for bit -- results, rasterised cells, grids."""
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    extra = [_mk_native(b) for b in base]

    def same(a, b):
        return (a.response == b.response and a.covariance == b.covariance and a.meta == b.meta and
                (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1]))

    for limit_kib in (0, 200):  # 200 KiB holds 7 scans of 1081 beams: the cache keeps starting over
        m, ref = ScanMatcher(), ScanMatcher()
        ref.debug_option(7, 1)
        if limit_kib:
            m.debug_option(8, limit_kib)
        plans = [
            ("single", nb), ("single", nb), ("move", 3), ("single", nb), ("single", nb[2:7]),
            ("batch", [nb, nb[:5], nb[5:], nb]), ("move", 0), ("move", 9), ("batch", [nb, nb[:5], nb[5:], nb]),
            ("batch", [nb[:4] + extra[:3], extra, nb]), ("move", 5), ("single", nb[::-1]),
            ("batch", [nb] * 9 + [extra] * 3), ("batch", [nb] * 9 + [extra] * 3),
        ]
        k = 0
        for kind, arg in plans:
            if kind == "move":
                p = nb[arg].corrected_pose
                k += 1
                nb[arg].corrected_pose = Transform(p.x + 0.013 * k, p.y - 0.007 * k, 0.0, p.euler[-1] + 0.011 * k)
                continue
            if kind == "single":
                a, b = m.match_scan(nq, arg, True, True), ref.match_scan(nq, arg, True, True)
                assert same(a, b), (limit_kib, kind)
                n_items = 1
            else:
                pa, pb = m.match_scan_batch(nq, arg, True, True)[0], ref.match_scan_batch(nq, arg, True, True)[0]
                assert all(same(a, b) for a, b in zip(pa, pb)), (limit_kib, kind)
                n_items = len(arg)
            for item in range(n_items):
                ca, _ = m.debug_cells(item)
                cb, _ = ref.debug_cells(item)
                assert np.array_equal(ca, cb)
                ga, _ = m.debug_grid(item)
                gb, _ = ref.debug_grid(item)
                assert np.array_equal(ga, gb)
        m.close()
        ref.close()
    # the yagpy path shares the kernel: same check on one golden case
    from tests.util import load_case
    c = load_case("small_dirty_rot")
    gq, gb = _mk_native(c["query"]), [_mk_native(b) for b in c["base"]]
    m, ref = ScanMatcher(c["cfg"], semantics="yagpy"), ScanMatcher(c["cfg"], semantics="yagpy")
    ref.debug_option(7, 1)
    for _ in range(2):
        a, b = m.match_scan(gq, gb, True, True), ref.match_scan(gq, gb, True, True)
        assert a.response == b.response and a.covariance == b.covariance
        assert np.array_equal(m.debug_grid()[0], ref.debug_grid()[0])


def test_chain_structure_from_scan_creation_equals_the_per_pose_computation():
    """The trigger chain of the valid-point filter is computed once per scan in the sensor frame (structure_kernel, with a
    guard band around the distance threshold) and used at every pose; debug option 24 = 0 recomputes it from the projected
    points at every pose, as Karto does.  Same cells, grids and results -- near the origin, rotated, 9 km out, single
    matches and batches, both semantics, and for a scan with a reading pair placed ON the threshold (whose structure is
    not trusted)."""
    import math
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    from tests.util import load_case
    q, base = cfg2_scans()
    # a pair of neighbouring readings exactly 0.1 m apart (law of cosines), inside a run of close readings
    r = np.array(base[3].ranges, dtype=np.float64)
    inc = base[3].angle_increment
    k = 500
    r[k - 3:k + 4] = 1.0
    r[k + 1] = math.cos(inc) + math.sqrt(0.01 - math.sin(inc) ** 2)
    base[3] = PlainScan(r, base[3].min_angle, inc, base[3].min_range, base[3].range_threshold,
                        (base[3].corrected_pose.x, base[3].corrected_pose.y, base[3].corrected_pose.euler[-1]))

    def same(a, b):
        return (a.response == b.response and a.covariance == b.covariance and a.meta == b.meta and
                (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1]))

    for sem in ("karto", "yagpy"):
        cfg = None if sem == "karto" else load_case("small_dirty_rot")["cfg"]
        m, ref = ScanMatcher(cfg, semantics=sem), ScanMatcher(cfg, semantics=sem)
        ref.debug_option(24, 0)
        for shift in ((0.0, 0.0, 0.0), (0.4, -0.3, 2.5), (9000.0, -8500.0, -1.1)):
            def moved(s):
                c = s.corrected_pose
                p = Transform(shift[0], shift[1], 0.0, shift[2]) + Transform(c.x, c.y, 0.0, c.euler[-1])
                return _mk_native(PlainScan(s.ranges, s.min_angle, s.angle_increment, s.min_range, s.range_threshold,
                                            (p.x, p.y, p.euler[-1])))
            nq, nb = moved(q), [moved(b) for b in base]
            a, b = m.match_scan(nq, nb, True, True), ref.match_scan(nq, nb, True, True)
            assert same(a, b), (sem, shift)
            assert np.array_equal(m.debug_cells(0)[0], ref.debug_cells(0)[0])
            assert np.array_equal(m.debug_grid(0)[0], ref.debug_grid(0)[0])
            chains = [nb, nb[:4], nb[3:], nb[::-1]] * 3
            pa, pb = m.match_scan_batch(nq, chains, True, True)[0], ref.match_scan_batch(nq, chains, True, True)[0]
            assert all(same(x, y) for x, y in zip(pa, pb)), (sem, shift)
            for item in (0, 1, 6, 11):
                assert np.array_equal(m.debug_cells(item)[0], ref.debug_cells(item)[0])
        m.close()
        ref.close()


def test_a_resident_batch_enqueued_again_replays_its_plan_only_while_nothing_moved():
    """`MatchBatch.run_async` on a slot that still holds the batch's call skips the per-scan host work (rebuilding the call,
    the point-cache lookups, the descriptor) as long as no scan was re-posed and no cache slot changed hands.  Results
    against a matcher that is built fresh for every enqueue: identical bits before and after base scans and the query
    move, with other calls (another batch, single matches, a dropped cache) in between, on the same and on other slots."""
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    extra = [_mk_native(b) for b in base[:4]]
    for k, s in enumerate(extra):
        p = s.corrected_pose
        s.corrected_pose = Transform(p.x + 0.05 * (k + 1), p.y - 0.03 * k, 0.0, p.euler[-1] + 0.02 * k)
    chains = [nb, nb[:5], nb[3:] + extra[:2], extra + nb[:3]] * 3
    other = [extra, nb[::-1]] * 5
    m = ScanMatcher()
    batch, batch2 = m.make_batch(nq, chains), m.make_batch(nq, other)

    def fresh():
        ref = ScanMatcher()
        out = ref.match_scan_batch(nq, chains, True, True)[0]
        ref.close()
        return [(r.response, r.covariance, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1]) for r in out]

    def run(slot):
        batch.run_async(True, True, slot=slot)
        per = batch.wait(slot)[0]
        return [(r.response, r.covariance, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1]) for r in per]

    def move(s, k):
        p = s.corrected_pose
        s.corrected_pose = Transform(p.x + 0.011 * k, p.y - 0.007 * k, 0.0, p.euler[-1] + 0.004 * k)

    want = fresh()
    for slot in (0, 0, 1, 0, 1, 1):            # first use, replay, another slot, replays
        assert run(slot) == want
    move(nb[4], 1)                             # a base scan moves: every slot must notice
    want = fresh()
    for slot in (0, 1, 0, 2):
        assert run(slot) == want
    batch2.run_async(True, True, slot=0)       # the slot is taken by another batch, then by this one again
    batch2.wait(0)
    assert run(0) == want and run(0) == want
    m.match_scan(nq, nb[:6], True, True)       # single matches do not disturb the slots
    assert run(1) == want
    m.debug_option(7, 2)                       # the point cache is dropped: a replayed plan would name dead slots
    assert run(1) == want and run(1) == want and run(0) == want
    move(nq, 1)                                # the query moves
    move(extra[1], 2)
    want = fresh()
    for slot in (0, 1, 0):
        assert run(slot) == want
    m.close()


def test_a_failed_call_leaves_no_unwritten_cache_entries():
    """Round-2 advisor finding: the point cache is updated on the host before the kernels that fill its entries are enqueued.
    A call that fails in between (here: an order-dependent smear on a chain of more readings than the select kernels
    take, refused after the cache was planned) must not leave those entries "current": the same scans in a chain that
    works must give what a fresh matcher gives."""
    from yag_slam_amd import synth
    from yag_slam_amd._capi import YmError
    from yag_slam_amd.scan_matching import ScanMatcher
    q, _ = cfg2_scans(range_threshold=12.0)
    scene = synth.Scene()
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 12.0, p)
    poses = [(2.0 + 0.02 * i, 3.0 + 0.01 * (i % 5), 0.01 * (i % 7)) for i in range(92)]   # 92 x 1081 = 99 452 readings
    chain = [_mk_native(mk(scene.scan_ranges(p, index=700 + i), p)) for i, p in enumerate(poses)]
    nq = _mk_native(q)
    cfg = dict(resolution=0.01, smear_deviation=0.1, range_threshold=12.0, search_size=0.3)
    m = ScanMatcher(cfg)
    with pytest.raises(YmError) as e:
        m.match_scan(nq, chain, True, True)
    assert e.value.code == -4
    got = m.match_scan(nq, chain[:6], True, True)
    ref = ScanMatcher(cfg).match_scan(nq, chain[:6], True, True)
    assert got.response == ref.response and got.covariance == ref.covariance and got.meta == ref.meta
    # and in a batch (the split prepare: points_kernel fills the entries)
    with pytest.raises(YmError):
        m.match_scan_batch(nq, [chain] * 8, True, True)
    per, _ = m.match_scan_batch(nq, [chain[10:16]] * 8, True, True)
    ref = ScanMatcher(cfg).match_scan(nq, chain[10:16], True, True)
    assert per[3].response == ref.response and per[3].covariance == ref.covariance


def test_order_dependent_smear_on_long_chains():
    """More than 12 288 readings in a chain at smear_deviation = 10 * resolution: Karto's "value already set" rule is
    evaluated with its tables in global memory instead of one CU's LDS.  Every grid byte must still equal the
    sequential oracle's; and the same chain through both kernels (option 10 forces the global one) gives the same grid."""
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd import synth
    scene = synth.Scene()
    q, base = cfg2_scans(range_threshold=12.0)
    mk = lambda r, p: PlainScan(r, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 12.0, p)
    poses = [(2.0 + 0.05 * i, 3.0 + 0.02 * (i % 5), 0.03 * (i % 7)) for i in range(26)]   # 26 x 1081 = 28 106 readings
    long_chain = [mk(scene.scan_ranges(p, index=400 + i), p) for i, p in enumerate(poses)]
    cfg = dict(resolution=0.01, smear_deviation=0.1, range_threshold=12.0, search_size=0.3)
    compare(cfg, q, long_chain, True, True)
    # both kernels on a chain either can hold
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    a, b = ScanMatcher(cfg), ScanMatcher(cfg)
    b.debug_option(10, 1)
    ra, rb = a.match_scan(nq, nb, True, True), b.match_scan(nq, nb, True, True)
    assert ra.response == rb.response and ra.covariance == rb.covariance
    assert np.array_equal(a.debug_grid()[0], b.debug_grid()[0])
    assert np.array_equal(a.debug_cells()[0], b.debug_cells()[0])


def test_merged_equal_offsets_are_exact():
    """On a coarse grid neighbouring beams fall into the same cell; the correlate kernel then merges runs of equal lookup
    offsets into one entry with a multiplicity (integer sums: exact).  Forced on (option 13 = 1) and off (= 2), on the
    loop config where it triggers by itself and on the fine default config where it never would: identical sum volumes
    and results, and equal to the oracle."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    short = [_mk_native(PlainScan(b.ranges[:300], b.min_angle, b.angle_increment, b.min_range, 20.0,
                                  (b.corrected_pose.x, b.corrected_pose.y, 0.0))) for b in base[:3]]
    qshort = _mk_native(PlainScan(q.ranges[:707], q.min_angle, q.angle_increment, q.min_range, 20.0, (3.0, 3.0, 0.0)))  # ragged last chunk
    for loop in (True, False):
        for query, chain in ((nq, nb), (qshort, short)):
            vols = []
            for mode in (1, 2, 0):
                m = ScanMatcher(None, loop=loop)
                m.debug_option(13, mode)
                r = m.match_scan(query, chain, False, True)
                vols.append((m.debug_sums(0, dims=r.meta["coarse_dims"]), r))
                # and inside a batch (the kernels with chunk-waves)
                per, _ = m.match_scan_batch(query, [chain] * 9, False, True)
                assert per[4].response == r.response and per[4].covariance == r.covariance
            assert np.array_equal(vols[0][0], vols[1][0]) and np.array_equal(vols[0][0], vols[2][0])
            assert vols[0][1].response == vols[1][1].response and vols[0][1].covariance == vols[1][1].covariance
    o = orc.Oracle(None, "karto", loop=True)
    ro = o.match_scan(q, base, False, False)
    m = ScanMatcher(None, loop=True)
    m.debug_option(13, 1)
    r = m.match_scan(nq, nb, False, False)
    assert np.array_equal(m.debug_sums(0, dims=r.meta["coarse_dims"]), o.sums(0)) and abs(r.response - ro["response"]) <= 1e-12


def test_region_correlate_equals_the_direct_kernel_and_the_oracle():
    """Batches of 8+ items gather their patches from LDS, region by region: correlate_region_kernel + bin_kernel on lattices
    up to 26 x 32, gather_kernel + the gbin kernels on other lattices up to 48 x 64 -- and, forced by option 14 = 4, here too;
    both sort a query's (beam, angle) pairs once per call.  The integer sum volumes of every item must be those of the
    direct correlate kernel (option 14 = 1) -- also through the per-cell path (= 2, what an item with a non-lattice
    hypothesis grid takes) and the "lists do not fit" path (= 3) -- for the default lattice, a lattice of one lane per row
    (nx = 13), ragged queries (1081 / 707 / 400 valid beams: a set of 16-bit sums written out mid-way, or none) and ragged
    and reversed chains; and equal to the oracle's.  Then the gather kernel with its work forced into other shapes: the
    angles of an item shared out over 1, 2, 3 and 21 blocks (option 17), 1 to 4 angles per wave (15), regions cut into chunks
    of 64 units (19), an LDS budget that makes the regions small (20), its per-cell kernel."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher as _ScanMatcher

    def ScanMatcher(*args, **kw):  # (the LDS correlates from 8 chains on, as production takes them from 64)
        m_ = _ScanMatcher(*args, **kw)
        m_.debug_option(28, 8)
        return m_
    experimental = _has_experimental_forms()
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    cut = lambda s, n: PlainScan(s.ranges[:n], s.min_angle, s.angle_increment, s.min_range, 20.0, (3.0, 3.0, 0.0))
    q707, q400 = cut(q, 707), cut(q, 400)
    narrow = {"search_size": 0.24}  # 13 x 13 x 21: one lane per lattice row
    chains = [nb, nb[:3], nb[2:9], nb[::-1], nb[4:5], nb[1:], nb[:7], nb[3:], nb[5:6]]  # (no empty chain: its response
    # expansion would be the last call, and debug_sums reads the last call)

    def same(a, b):
        for x, y in zip(a[1], b[1]):
            assert x.response == y.response and x.covariance == y.covariance and x.meta == y.meta
            assert (x.best_pose.x, x.best_pose.y, x.best_pose.euler[-1]) == (y.best_pose.x, y.best_pose.y, y.best_pose.euler[-1])
        assert a[2] == b[2]

    for cfg, queries in ((None, (q, q707, q400)), (narrow, (q,))):
        for query in queries:
            nquery = _mk_native(query)
            vols = {}
            for mode in (0, 1, 2, 3, 4):
                m = ScanMatcher(cfg)
                m.debug_option(12, 1)  # keep the integer sums of batches
                m.debug_option(14, mode)
                per, best = m.match_scan_batch(nquery, chains, True, True)
                dims = per[0].meta["coarse_dims"]
                vols[mode] = ([m.debug_sums(0, item=i, dims=dims) for i in range(len(chains))], per, best)
            for mode in (0, 2, 3, 4):
                for i in range(len(chains)):
                    assert np.array_equal(vols[mode][0][i], vols[1][0][i]), (mode, i)
                same(vols[mode], vols[1])
            assert vols[0][0][0].any()
            # the region correlate's other forms: wave-specialised (option 32 = 2: gather waves + loader waves, persistent blocks) and
            # one block per item with the item's sums in LDS (= 3: what large batches take), with the lists that fit and without
            # and the pooled form at two blocks per item (= 4: a region's patches dealt evenly over twelve waves, 16-bit sums in LDS)
            # round 5: sixteen waves per block, two or three per angle (option 32 = 5; 43 = region height: form 80 / 100 / 128)
            for form, irregular in ((2, 0), (3, 0), (3, 2), (3, 3), (4, 0), (4, 2), (4, 3), (80, 0), (100, 0), (128, 0), (128, 2), (100, 3)) if experimental else ():
                m = ScanMatcher(cfg)
                m.debug_option(12, 1)
                if form >= 80:
                    m.debug_option(43, form)
                m.debug_option(32, 5 if form >= 80 else form)
                if irregular:
                    m.debug_option(14, irregular)
                per, best = m.match_scan_batch(nquery, chains, True, True)
                other = ([m.debug_sums(0, item=i, dims=per[0].meta["coarse_dims"]) for i in range(len(chains))], per, best)
                for i in range(len(chains)):
                    assert np.array_equal(other[0][i], vols[1][0][i]), (form, irregular, i)
                same(other, vols[1])
            # without the kept sums (the production form: the region correlate scores its sums itself; option 21 = 2 leaves
            # that to the score kernel), through either kernel
            for opts in ({}, {21: 2}, {14: 4}, {32: 2}, {32: 2, 21: 2}, {32: 3}, {32: 3, 21: 2}, {32: 3, 14: 2}, {32: 4}, {32: 4, 21: 2}, {32: 4, 14: 2}, {32: 4, 39: 1}):
                if 32 in opts and not experimental:
                    continue
                if opts.get(32) == 4 and len(opts) == 1:  # (rides along: the sixteen-wave form with its three region heights)
                    for h in (80, 100, 128):
                        m = ScanMatcher(cfg)
                        m.debug_option(43, h)
                        m.debug_option(32, 5)
                        same((None,) + tuple(m.match_scan_batch(nquery, chains, True, True)), vols[1])
                m = ScanMatcher(cfg)
                for k, v in opts.items():
                    m.debug_option(k, v)
                res = m.match_scan_batch(nquery, chains, True, True)
                same((None,) + tuple(res), vols[1])
            o = orc.Oracle(cfg, "karto")
            o.match_scan(query, base, True, True)
            assert np.array_equal(vols[0][0][0], o.sums(0))
    ref = None
    for opts in ({}, {17: 1}, {17: 2}, {17: 3}, {17: 21}, {15: 1, 17: 2}, {15: 2}, {15: 4}, {19: 64}, {20: 30000}, {20: 20000, 19: 64, 17: 2}):
        m = ScanMatcher()
        m.debug_option(12, 1)
        m.debug_option(14, 4)
        for k, v in opts.items():
            m.debug_option(k, v)
        per, best = m.match_scan_batch(nq, chains, True, True)
        dims = per[0].meta["coarse_dims"]
        got = ([m.debug_sums(0, item=i, dims=dims) for i in range(len(chains))], per, best)
        if ref is None:
            ref = got
            continue
        for a, b in zip(got[0], ref[0]):
            assert np.array_equal(a, b), opts
        same(got, ref)


def test_raster_grid_shorter_than_the_tile_list():
    """On batches the raster launches as many blocks per item as the longest tile list of the matcher's previous call
    (+ 1/8); entries beyond the grid are rasterised by a second, small launch that walks them.  Forced grids of 1, 7 and
    40 blocks per item (option 16) and the free-running matcher (first call: one block per tile; later calls: by the
    hint) must produce the same windows and results -- as must every size of the per-tile hit lists."""
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    chains = [nb, nb[:3], nb[2:9], nb[::-1], nb[4:5], nb[1:], nb[:7], nb[3:], nb[5:6]] * 6  # (tile lists: from 48 items on)
    ref = None
    for gx, hits in ((0, 0), (0, 0), (1, 0), (7, 0), (40, 0), (0, -1), (0, 1), (0, 4)):
        # (hits: room in the per-tile lists of chunk boxes the tiles kernel builds, option 18; -1 = none, 1 and 4 per tile =
        # the lists of most / some items do not fit and their raster blocks scan the boxes themselves)
        m = ScanMatcher() if ref is None or gx or hits else m
        m.debug_option(16, gx)
        m.debug_option(18, hits)
        per, best = m.match_scan_batch(nq, chains, True, True)
        grids = [m.debug_grid(i)[0] for i in (0, 4, 8)]
        if ref is None:
            ref = (per, best, grids)
            continue
        for a, b in zip(per, ref[0]):
            assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta
        assert best == ref[1]
        for g, r in zip(grids, ref[2]):
            assert np.array_equal(g, r)


@pytest.mark.parametrize("gx", [0, 100000])
def test_raster_sub_block_knowledge_is_invisible(gx):
    """The raster keeps, per 8 x 8 sub-block of a window tile, whether the window memory holds zeros there, and does not store
    a sub-block again that is zero now and was zero before.  One matcher sees the same chains at a sequence of poses (walls
    move across sub-blocks and tiles, come back, disappear): after every call its windows -- single match and batch of 50,
    work lists and hit slots -- are byte for byte those of a matcher that has never seen anything else.  (gx = 100000: one
    block per listed tile in ONE launch, the most blocks in flight -- how a race between a block's waves on the tile's flag
    first showed.)"""
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    scene = synth.Scene()
    base_poses, q_truth, q_prior = synth.single_match_poses()
    ranges = [scene.scan_ranges(p, index=i) for i, p in enumerate(base_poses)]
    q_ranges = scene.scan_ranges(q_truth, index=10)
    shifts = [(0.0, 0.0, 0.0), (0.13, -0.07, 0.02), (0.0, 0.0, 0.0), (-0.31, 0.22, -0.05), (0.02, 0.01, 0.0), (1.7, -0.9, 0.3), (0.0, 0.0, 0.0)]
    m = ScanMatcher({"use_response_expansion": False})
    m.debug_option(16, gx)
    for k, (dx, dy, dt) in enumerate(shifts):
        # (the chain moves, the query stays: the window is anchored at the query)
        base = [synth.resident_scan(r, (p[0] + dx, p[1] + dy, p[2] + dt)) for r, p in zip(ranges, base_poses)]
        query = synth.resident_scan(q_ranges, q_prior)
        chains = [base[:max(1, len(base) - (c % 4))] for c in range(50)]
        fresh = ScanMatcher({"use_response_expansion": False})
        for mm in (m, fresh):
            mm.match_scan(query, base, True, True)
        g, _ = m.debug_grid()
        gf, _ = fresh.debug_grid()
        assert np.array_equal(g, gf), "call %d, single match: %d cells differ" % (k, int((g != gf).sum()))
        pm, _ = m.match_scan_batch(query, chains, True, True)
        pf, _ = fresh.match_scan_batch(query, chains, True, True)
        for i in range(50):
            g, _ = m.debug_grid(item=i)
            gf, _ = fresh.debug_grid(item=i)
            assert np.array_equal(g, gf), "call %d, item %d: %d cells differ" % (k, i, int((g != gf).sum()))
        assert [r.response for r in pm] == [r.response for r in pf]
        fresh.close()
    m.close()


def test_window_only_calls_and_plane_calls_on_one_matcher():
    """A batch of 64+ items on the default lattice stages its regions from the row-major window and its raster writes no
    column planes; single matches and small batches correlate from the planes.  One matcher takes both kinds of call in
    turn, with the chains moving in between (so that planes left behind by an earlier call would be wrong): every call's
    responses, poses and covariances are those of a matcher that has seen nothing else, and the integer sum volumes of
    the large batch equal those of the same batch with the planes kept (debug option 39)."""
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    scene = synth.Scene()
    base_poses, q_truth, q_prior = synth.single_match_poses()
    ranges = [scene.scan_ranges(p, index=i) for i, p in enumerate(base_poses)]
    query = synth.resident_scan(scene.scan_ranges(q_truth, index=10), q_prior)
    m = ScanMatcher({"use_response_expansion": False})
    m.debug_option(12, 1)  # (keep the integer sums)

    def same(a, b):
        assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta
        assert (a.best_pose.x, a.best_pose.y, a.best_pose.euler[-1]) == (b.best_pose.x, b.best_pose.y, b.best_pose.euler[-1])
    # (round 5: which items' planes lag behind is kept per item -- a plane call of 12 items after a window-only call of 80 brings
    #  items 0 .. 11 up to date and leaves 12 .. 79 behind, the plane call of 30 items after it must still refresh 12 .. 29)
    for k, (n, dx, dy, dt) in enumerate([(80, 0.0, 0.0, 0.0), (1, 0.21, -0.13, 0.03), (80, 0.21, -0.13, 0.03), (50, -0.4, 0.3, -0.06),
                                         (1, -0.4, 0.3, -0.06), (96, 0.05, 0.02, 0.01), (1, 0.0, 0.0, 0.0),
                                         (80, 0.3, 0.1, 0.02), (12, -0.2, 0.25, -0.04), (30, 0.12, -0.3, 0.05), (64, 0.12, -0.3, 0.05),
                                         (30, -0.1, -0.1, 0.0), (1, 0.02, 0.0, 0.0)]):
        base = [synth.resident_scan(r, (p[0] + dx, p[1] + dy, p[2] + dt)) for r, p in zip(ranges, base_poses)]
        fresh = ScanMatcher({"use_response_expansion": False})
        fresh.debug_option(12, 1)
        fresh.debug_option(39, 1)  # (the fresh matcher always keeps the planes)
        if n == 1:
            same(m.match_scan(query, base, True, True), fresh.match_scan(query, base, True, True))
        else:
            chains = [base[:max(1, len(base) - (c % 4))] for c in range(n)]
            pm, bm = m.match_scan_batch(query, chains, True, True)
            pf, bf = fresh.match_scan_batch(query, chains, True, True)
            assert bm == bf
            for a, b in zip(pm, pf):
                same(a, b)
            dims = pm[0].meta["coarse_dims"]
            for i in (0, 3, n - 1):
                assert np.array_equal(m.debug_sums(0, item=i, dims=dims), fresh.debug_sums(0, item=i, dims=dims)), "call %d item %d" % (k, i)
        fresh.close()
    m.close()


@pytest.mark.parametrize("sigma_cells", [1.0, 3.0, 6.0, 7.0])
def test_raster_row_tables_are_invisible(sigma_cells):
    """The raster's row pass finds an 8-cell group's distances to the nearest occupied cell of its row in tables indexed by
    seven bitmap bits at a time (kernel half widths up to 12: smear deviations up to 6 cells; beyond that by 64-bit bit scans).
    The same windows, byte for byte, as the bit scans (debug option 37) and as the oracle's max-stamp -- single match
    (one block per tile) and a batch of 50 (work lists, hit slots) -- for kernel halves 2, 6, 12 and 14."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans()
    cfg = {"smear_deviation": 0.01 * sigma_cells, "use_response_expansion": False}
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    chains = [nb[:max(1, len(nb) - (c % 4))] for c in range(50)]
    o = orc.Oracle(cfg, "karto")
    o.match_scan(q, base, True, True)
    og, _ = o.grid_u8()
    seen = None
    for scans in (0, 1, 0):
        m = ScanMatcher(cfg)
        m.debug_option(37, scans)
        r = m.match_scan(nq, nb, True, True)
        g, info = m.debug_grid()
        per, _ = m.match_scan_batch(nq, chains, True, True)
        gb = [m.debug_grid(item=i)[0] for i in (0, 3, 49)]
        key = (r.response, r.covariance, tuple(p.response for p in per))
        if seen is None:
            seen = (key, g.copy(), [x.copy() for x in gb])
            sub = og[info.origin_y:info.origin_y + info.height, info.origin_x:info.origin_x + info.width]
            assert np.array_equal(g, sub), "grid window differs from the oracle's in %d cells" % int((g != sub).sum())
            assert np.array_equal(gb[0], g)
        else:
            assert key == seen[0]
            assert np.array_equal(g, seen[1])
            for x, y in zip(gb, seen[2]):
                assert np.array_equal(x, y)
        m.close()


@pytest.mark.parametrize("nx", [3, 7, 13, 16, 17, 26, 27, 32, 33, 41, 47])
def test_region_correlate_lattice_widths(nx):
    """The gather correlate gives a lane 16 x-adjacent hypotheses of one lattice row, a wave up to 32 rows of two such
    segments; lattices past 32 rows or two segments take two or three waves per angle: every shape from one short lane to
    three full segments over 47 rows, sums against the direct kernel, results against single calls."""
    from yag_slam_amd.scan_matching import ScanMatcher as _ScanMatcher

    def ScanMatcher(*args, **kw):  # (the LDS correlates from 8 chains on, as production takes them from 64)
        m_ = _ScanMatcher(*args, **kw)
        m_.debug_option(28, 8)
        return m_
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    cfg = {"search_size": (nx - 1) * 0.02, "coarse_search_angle_offset": 0.07 if (nx < 13 or nx > 33) else 0.349}
    chains = [nb, nb[:4], nb[3:], nb[::-1], nb[2:7], nb[5:], nb[:8], nb[1:9]]
    vols = []
    for mode in (0, 1):
        m = ScanMatcher(cfg)
        m.debug_option(12, 1)
        m.debug_option(14, mode)
        per, best = m.match_scan_batch(nq, chains, True, True)
        dims = per[0].meta["coarse_dims"]
        assert dims[0] == nx and dims[1] == nx
        vols.append([m.debug_sums(0, item=i, dims=dims) for i in range(len(chains))])
    for a, b in zip(*vols):
        assert np.array_equal(a, b)
    # ... and the first two chains' volumes against the oracle itself (not only through the direct kernel)
    from oracle import oracle as orc
    for i in (0, 1):
        o = orc.Oracle(cfg, "karto")
        ro = o.match_scan(q, [base[j] for j in ([range(10), range(4)][i])], True, True)
        if ro["expansions"] == 0 and per[i].meta["expansions"] == 0:
            assert np.array_equal(vols[1][i], o.sums(0)), "gather correlate differs from the oracle at width %d, chain %d" % (nx, i)
    m = ScanMatcher(cfg)
    per, best = m.match_scan_batch(nq, chains, True, True)
    singles = [m.match_scan(nq, ch, True, True) for ch in chains]
    for a, b in zip(per, singles):
        assert a.response == b.response and a.covariance == b.covariance and a.meta == b.meta
    assert best == int(np.argmax([s.response for s in singles]))


def test_region_correlate_on_the_loop_config_with_multiplicities():
    """BASELINE configs[3]'s lattice (41 x 41 x 21 on 5 cm cells): two waves per angle, and neighbouring beams fall into one
    cell, so the lists hold patches with multiplicities 1..4 and more (runs of 5+ beams are cut).  Sum volumes of a batch
    against the direct kernel with and without its own run merging, against the oracle, and with the gather's work forced
    into other shapes; a short-range query (runs of 10+ beams per cell) as well."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher as _ScanMatcher

    def ScanMatcher(*args, **kw):  # (the LDS correlates from 8 chains on, as production takes them from 64)
        m_ = _ScanMatcher(*args, **kw)
        m_.debug_option(28, 8)
        return m_
    q, base = cfg2_scans()
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    near = PlainScan(np.minimum(q.ranges, 0.6), q.min_angle, q.angle_increment, q.min_range, 20.0, (3.0, 3.0, 0.0))
    chains = [nb, nb[:3], nb[2:9], nb[::-1], nb[4:5], nb[1:], nb[:7], nb[3:]]
    for query in (q, near):
        nquery = _mk_native(query)
        ref = None
        for opts in ({14: 1, 13: 2}, {14: 1, 13: 1}, {}, {14: 2}, {17: 2}, {17: 3, 15: 1}, {19: 64}, {20: 30000}):
            m = ScanMatcher(None, loop=True)
            m.debug_option(12, 1)
            for k, v in opts.items():
                m.debug_option(k, v)
            per, best = m.match_scan_batch(nquery, chains, False, False)
            dims = per[0].meta["coarse_dims"]
            assert tuple(dims) == (41, 41, 21)
            got = ([m.debug_sums(0, item=i, dims=dims) for i in range(len(chains))], per, best)
            if ref is None:
                ref = got
                continue
            for a, b in zip(got[0], ref[0]):
                assert np.array_equal(a, b), opts
            for x, y in zip(got[1], ref[1]):
                assert x.response == y.response and x.covariance == y.covariance and x.meta == y.meta, opts
            assert got[2] == ref[2]
        o = orc.Oracle(None, "karto", loop=True)
        o.match_scan(query, base, False, False)
        assert np.array_equal(ref[0][0], o.sums(0))


def test_scan_pool_staging_and_recycled_blocks_are_invisible():
    """ym_scan_create uploads through a ring of 64 pinned staging slots with one launch and no synchronisation, and the
    blocks of destroyed scans are handed out again after the pool's next device-wide synchronisation: scans created
    faster than the ring turns, used at once, and scans living in blocks that held other readings before must match
    exactly like the first time, and the structure's info must be the same whenever it is read."""
    from yag_slam_amd import _capi, synth
    from yag_slam_amd.scan_matching import ScanMatcher
    L = _capi.lib()
    truth, scans = synth.trajectory_scans(260)
    m = ScanMatcher()

    def run(seq):
        out = []
        for k in range(20, len(seq), 24):
            r = m.match_scan(seq[k], seq[k - 20:k], True, True)
            out.append((r.response, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1], r.covariance, r.meta))
        return out

    for s in scans:       # 260 creations back to back: the ring wraps four times before anything waits
        s.native(0)
    first = run(scans)    # ... and the first use follows at once
    trusted = [L.ym_scan_structure_trusted(s.native(0), 0) for s in scans]
    for s in scans:
        s._release()
    # other readings go through the freed blocks ...
    other = synth.trajectory_scans(200, scene=synth.Scene(seed=7))[1]
    for s in other:
        s.native(0)
    for s in other:
        s._release()
    # ... and the first set comes back, in another order of creation
    for s in reversed(scans):
        s.native(0)
    assert [L.ym_scan_structure_trusted(s.native(0), 0) for s in scans] == trusted
    assert run(scans) == first


def test_bulk_scan_creation_equals_per_scan_creation():
    """ym_scans_create (one pool transaction, one upload, one launch per 2048 scans) must leave behind exactly what one ym_scan_create
    per scan does: the same structure info, the same matches -- single calls and the region-staged batch path -- for ordinary scans,
    dirty ones, scans of other lengths, a scan without a single valid reading, an empty scan; more scans than one chunk holds; blocks
    recycled between the two ways of creating; the array form (models.ScanBlock) and the bulk destroy."""
    from yag_slam_amd import _capi, synth
    from yag_slam_amd.models import LocalizedRangeScan, ScanBlock, native_many
    from yag_slam_amd.scan_matching import MatchBatch, ScanMatcher
    L = _capi.lib()
    scene = synth.Scene()
    truth, scans = synth.trajectory_scans(130)
    odd = [synth.resident_scan(scene.scan_ranges(truth[3], index=900, dirty=True), truth[3]),
           LocalizedRangeScan(scans[5].ranges[:300], synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, 20.0, *truth[5]),
           LocalizedRangeScan(np.full(1081, np.nan), synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, 20.0, *truth[6]),
           LocalizedRangeScan(np.zeros(0), synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, 20.0, *truth[7]),
           LocalizedRangeScan(np.full(5000, 4.0), -3.0, 3.0, 6.0 / 4999, synth.MIN_RANGE, synth.MAX_RANGE, 20.0, *truth[8])]
    for s, t in zip(scans, truth):  # (matching needs the scans where they were taken)
        s.corrected_pose = type(s.corrected_pose)(t[0], t[1], 0.0, t[2])
    everything = scans + odd
    m = ScanMatcher()

    def observe(seq):
        out = [tuple(L.ym_scan_structure_trusted(s.native(0), sem) for sem in (0, 1)) for s in seq]
        for k in range(12, 120, 27):
            r = m.match_scan(seq[k], seq[k - 10:k], True, True)
            out.append((r.response, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1], r.covariance, r.meta))
        # ragged chains with the odd scans in them, as a batch on the region-staged path (one query each: ym_pairs_create)
        m.debug_option(28, 8)
        qs = [seq[20 + i] for i in range(9)]
        chains = [seq[10 + i:20 + i] for i in range(5)] + [[seq[130 + j]] + seq[15:18] for j in range(4)]
        per = m.match_pairs(qs, chains, True, True)
        m.debug_option(28, 0)
        out.append([(p.response, p.best_pose.x, p.best_pose.y, p.best_pose.euler[-1], p.covariance) for p in per])
        return out

    for s in everything:
        s.native(0)
    one_by_one = observe(everything)
    for s in everything:
        s._release()
    native_many(everything, 0)  # the same scans, in the blocks the first set has just given back
    assert all(s._native for s in everything)
    assert observe(everything) == one_by_one
    # more scans than one chunk (2048) holds, through the array form: every structure word as the per-scan creation reports it
    n = 2500
    ranges = np.stack([scans[i % 130].ranges for i in range(n)])
    poses = np.array([truth[i % 130] for i in range(n)])
    sensor = (synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, synth.RANGE_THRESHOLD)
    blk = ScanBlock(ranges, poses, sensor, device=0)
    assert len(blk) == n and np.all(blk.handles != 0)
    for i in (0, 1, 129, 2047, 2048, 2049, n - 1):
        assert tuple(L.ym_scan_structure_trusted(int(blk.handles[i]), sem) for sem in (0, 1)) == one_by_one[i % 130][:2]
    # ... and matches: items of the block against chains of the block
    qh = blk.handles[[2060 + k for k in range(8)]]            # = scans 12 .. 19 of the trajectory
    sh = np.concatenate([blk.handles[2050 + k:2060 + k] for k in range(8)])
    hb = MatchBatch.from_handles(m, qh, sh, np.arange(9) * 10)
    hb.run_async(True, True, slot=0)
    per, _, _ = hb.wait(0)
    for k in range(8):
        r = m.match_scan(everything[(2060 + k) % 130], [everything[(2050 + k + j) % 130] for j in range(10)], True, True)
        assert (r.response, r.covariance) == (per[k].response, per[k].covariance), k
    hb.close()
    blk.release()
    assert len(blk) == 0


@pytest.mark.parametrize("seed", [0, 3, 5, 8, 11, 14])
def test_raster_tile_height_is_invisible(seed):
    """Large batches over large windows rasterise in 64 x 64 tiles, everything else in 64 x 32 (a host decision per call):
    forced either way on the same matcher in turn (debug option 30: the tile-zero flags and dirty rectangles of one tiling
    must not leak into the other), single matches and batches give the same window bytes and results, and the tall
    tiling's bytes are the oracle's."""
    from oracle import oracle as orc
    from yag_slam_amd.scan_matching import ScanMatcher
    cfg, query, base, penalty, fine, _ = _random_case(seed)
    o = orc.Oracle(cfg, "karto")
    ro = o.match_scan(query, base, penalty, fine)
    m = ScanMatcher(cfg)
    nq, nb = _mk_native(query), [_mk_native(b) for b in base]
    chains = [nb[:max(1, len(nb) - (c % 3))] for c in range(50)]  # (50 items: tile work lists + hit lists)
    seen = {}
    for th in (64, 32, 64, 0):
        m.debug_option(30, th)
        r = m.match_scan(nq, nb, penalty, fine)
        g, info = m.debug_grid()
        per, _ = m.match_scan_batch(nq, chains, penalty, fine)
        gb, _ = m.debug_grid(item=49)
        key = (r.response, r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1], r.covariance, r.meta,
               tuple((p.response, p.best_pose.x, p.best_pose.y, p.best_pose.euler[-1]) for p in per))
        if seen:
            assert key == seen["key"], "results differ at tile height %d" % th
            assert np.array_equal(g, seen["g"]) and np.array_equal(gb, seen["gb"]), "window bytes differ at tile height %d" % th
        else:
            seen = {"key": key, "g": g.copy(), "gb": gb.copy()}
            if ro["n_query_points"] > 0 and ro["expansions"] == 0 and r.meta["expansions"] == 0:
                og, oinfo = o.grid_u8()
                sub = og[info.origin_y:info.origin_y + info.height, info.origin_x:info.origin_x + info.width]
                assert np.array_equal(g, sub), "tall tiles: grid window differs from the oracle's in %d cells" % int((g != sub).sum())
    assert abs(r.response - ro["response"]) <= 1e-12


def test_a_query_matched_right_after_its_creation_is_read_from_its_staging_slot():
    """ym_scan_create returns before the scan's upload has completed; a synchronous match whose QUERY is such a scan reads
    the readings from the pinned staging slot instead of waiting (debug option 31 = 0: it waits).  Same results either
    way, the scan works as a base scan afterwards, and a sequence of create-match steps equals the one with waits."""
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.scan_matching import ScanMatcher

    def run(staged):
        truth, scans = synth.trajectory_scans(120)
        m = ScanMatcher()
        m.debug_option(31, 1 if staged else 0)
        mapper = SequentialMapper(m)
        out = []
        for s in scans:
            s.native(0)                      # created now ...
            r = mapper.process_scan(s)       # ... and matched at once
            p = s.corrected_pose
            out.append((p.x, p.y, p.euler[-1], None if r is None else r.response))
        return out

    assert run(True) == run(False)


def test_scan_pool_under_threads():
    """Four threads create scans (one pool, one ring of staging slots per device), match each at once against their own
    running chain on their own matcher -- so queries are read from staging slots while other threads turn the ring -- and
    retire the oldest: every thread gets the poses of the same run done alone."""
    import threading
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.scan_matching import ScanMatcher

    def run(n):
        truth, scans = synth.trajectory_scans(n)
        mapper = SequentialMapper(ScanMatcher())
        out = []
        for i, s in enumerate(scans):
            s.native(0)
            mapper.process_scan(s)
            p = s.corrected_pose
            out.append((p.x, p.y, p.euler[-1]))
            if i >= 80:
                scans[i - 80]._release()
        return out

    want = run(240)
    got, errors = {}, []

    def work(t):
        try:
            got[t] = run(240)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(got[t] == want for t in range(4))


def test_large_lattice_fine_stage_against_the_oracle():
    """configs[4]'s lattice (201 x 201 x 46) on single matches: the fine stage for lattices beyond 2048 cells keeps sixteen block
    maxima per thread in registers and 32 loads of the covariance block in flight (`fine_kernel<true>`): response, pose and
    covariance as the oracle's, on the full chain, a short one and one scan."""
    cfg = dict(search_size=2.0, resolution=0.005, coarse_search_angle_offset=0.785)
    q, base = cfg2_scans()
    for chain in (base, base[:4], base[7:8]):
        compare(cfg, q, chain, True, True)


def test_order_dependent_smear_split_form():
    """On up to eight items the "value already set" rule runs split: hash and earlier-neighbour search as launches over all
    points (tables in global memory, left zeroed), the chain of decisions in one block per item.  The surviving cells and the
    grid equal the one-block kernel's (debug option 41 = 0) on single matches and on a batch of five ragged chains, call
    after call on one matcher (the tables must come back clean), and the oracle's sequential rule."""
    from yag_slam_amd.scan_matching import ScanMatcher
    q, base = cfg2_scans(range_threshold=12.0)
    nq, nb = _mk_native(q), [_mk_native(b) for b in base]
    cfg = dict(resolution=0.005, smear_deviation=0.05, range_threshold=12.0)
    a, b = ScanMatcher(cfg), ScanMatcher(cfg)
    b.debug_option(41, 0)
    for chain in (nb, nb[:3], nb[4:], nb[::-1], nb[2:3], nb):
        ra, rb = a.match_scan(nq, chain, True, True), b.match_scan(nq, chain, True, True)
        assert ra.response == rb.response and ra.covariance == rb.covariance
        assert np.array_equal(a.debug_cells()[0], b.debug_cells()[0])
        assert np.array_equal(a.debug_grid()[0], b.debug_grid()[0])
    chains = [nb[:4], nb[2:6], nb[:2], nb[5:], nb[:4][::-1]]
    for _ in range(2):
        pa, _ = a.match_scan_batch(nq, chains, True, True)
        pb, _ = b.match_scan_batch(nq, chains, True, True)
        for x, y, ch in zip(pa, pb, chains):
            assert x.response == y.response and x.covariance == y.covariance
            z = a.match_scan(nq, ch, True, True)
            assert x.response == z.response and x.covariance == z.covariance
    a.close(); b.close()
    compare(cfg, q, base[3:8], True, True)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_order_dependent_smear_random_chains(seed):
    """Random chains (a random subset of the base scans in random order, poses jittered by up to 3 cm / 0.02 rad, one scan
    repeated so that whole walls coincide cell for cell) under smear_deviation = 10 * resolution: window, sums, response, pose
    and covariance of the split select rule against the oracle's sequential "value already set" rule."""
    rng = np.random.default_rng(seed)
    q, base = cfg2_scans(range_threshold=12.0)
    order = list(rng.permutation(len(base))[: int(rng.integers(2, 7))])
    order.append(order[0])  # (the same scan again: every one of its cells is taken already)
    chain = []
    for i in order:
        b = base[i]
        p = b.corrected_pose
        chain.append(PlainScan(b.ranges, b.min_angle, b.angle_increment, b.min_range, 12.0,
                               (p.x + rng.uniform(-0.03, 0.03), p.y + rng.uniform(-0.03, 0.03), p.euler[-1] + rng.uniform(-0.02, 0.02))))
    chain[-1] = PlainScan(base[order[0]].ranges, base[order[0]].min_angle, base[order[0]].angle_increment, base[order[0]].min_range, 12.0,
                          (chain[0].corrected_pose.x, chain[0].corrected_pose.y, chain[0].corrected_pose.euler[-1]))
    compare(dict(resolution=0.005, smear_deviation=0.05, range_threshold=12.0, search_size=0.2), q, chain, True, True)


def test_two_matchers_on_two_streams_with_pose_writes_between_their_enqueues():
    """bench.py's lanes in small: two matchers, each on its own stream, enqueue resident batches alternately WITHOUT waiting for
    each other; the batches share scan objects (one scan pool, one global pose epoch, a point cache per matcher) and poses are
    written between the enqueues.  Every enqueue must give what a third matcher gives for the same batch at the pose state of
    the moment it was enqueued, bit for bit."""
    import torch
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    scene = synth.Scene()
    q, _ = synth.single_match_scans(scene)
    base_poses, _, _ = synth.single_match_poses()
    exact = [scene.cast(*p) for p in base_poses]
    chains = []
    for c in range(96):
        rng = np.random.default_rng(7000 + c)
        chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    A, B = chains[:64], chains[32:]  # chains 32..63 belong to both batches
    rounds = 4

    def write(r, half):
        # a base scan of a shared chain, the query, and a scan of a chain only the OTHER batch holds
        for s, d in ((chains[40 + r][3], 0.004 * (r + 1)), (q, 0.002 * (r + 1)), (chains[70 + r if half == 0 else 5 + r][6], -0.003 * (r + 1))):
            p = s.corrected_pose
            s.corrected_pose = Transform(p.x + d, p.y - 0.5 * d, 0.0, p.euler[-1] + 0.1 * d)

    def snapshot():
        return [(s, s.corrected_pose) for ch in chains for s in ch] + [(q, q.corrected_pose)]
    start = snapshot()
    key = lambda per: [(p.response, tuple(map(tuple, p.covariance)), p.best_pose.x, p.best_pose.y, p.best_pose.euler[-1]) for p in per]
    # pass 1: the reference, one synchronous matcher replaying the sequence of enqueues and writes
    m0 = ScanMatcher()
    want = []
    for r in range(rounds):
        want.append(key(m0.match_scan_batch(q, A, True, True)[0]))
        write(r, 0)
        want.append(key(m0.match_scan_batch(q, B, True, True)[0]))
        write(r, 1)
    for s, p in start:
        s.corrected_pose = p
    # pass 2: two lanes, nothing waits until everything is enqueued
    m1, m2 = ScanMatcher(), ScanMatcher()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    m1.set_stream(s1.cuda_stream)
    m2.set_stream(s2.cuda_stream)
    b1, b2 = m1.make_batch(q, A), m2.make_batch(q, B)
    for r in range(rounds):
        b1.run_async(True, True, slot=r)
        write(r, 0)
        b2.run_async(True, True, slot=r)
        write(r, 1)
    got = []
    for r in range(rounds):
        got.append(key(b1.wait(r)[0]))
        got.append(key(b2.wait(r)[0]))
    torch.cuda.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, "enqueue %d (lane %d, round %d) differs from the single-matcher result" % (i, i % 2, i // 2)
    assert want[0] != want[2]  # (the writes did change results: the test compares something)


def test_pair_lists_kept_from_call_to_call_are_invisible():
    """A single-query batch whose query, pose, window and lattice equal those of the matcher's last list build does not build its
    pair lists again (round 5).  One matcher through a sequence that hits and misses that cache -- the same query against other
    chains, the query re-posed, another query, a multi-query call in between, a smaller batch that takes the direct kernel, the
    first query again -- must give what a matcher that always builds them (option 45 = 0) gives, bit for bit."""
    from yag_slam_amd import synth
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    scene = synth.Scene()
    q, _ = synth.single_match_scans(scene)
    base_poses, q_truth, q_prior = synth.single_match_poses()
    q2 = synth.resident_scan(scene.scan_ranges((3.02, 2.97, -0.03), index=77), (3.0, 3.0, 0.0))
    exact = [scene.cast(*p) for p in base_poses]
    chains = []
    for c in range(130):
        rng = np.random.default_rng(9000 + c)
        chains.append([synth.resident_scan(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    m, ref = ScanMatcher(), ScanMatcher()
    ref.debug_option(45, 0)
    key = lambda per: [(p.response, tuple(map(tuple, p.covariance)), p.best_pose.x, p.best_pose.y, p.best_pose.euler[-1]) for p in per]

    def both(query, chs, pairs=False):
        if pairs:
            a, b_ = m.match_pairs(query, chs, True, True), ref.match_pairs(query, chs, True, True)
        else:
            a, b_ = m.match_scan_batch(query, chs, True, True)[0], ref.match_scan_batch(query, chs, True, True)[0]
        assert key(a) == key(b_)
        return key(a)
    hits = lambda: m.debug_counters()["list_cache_hits"]
    r1 = both(q, chains[:64])
    assert hits() == 0
    r2 = both(q, chains[64:128])         # same query, other chains: the lists are in place
    assert r1 != r2
    assert hits() == 1                   # (the cache is really hit: the key compares equal, padding bytes and all)
    both(q, chains[:64])
    assert hits() == 2
    p = q.corrected_pose
    q.corrected_pose = Transform(p.x + 0.013, p.y - 0.004, 0.0, p.euler[-1] + 0.006)
    r3 = both(q, chains[:64])            # re-posed: built again
    assert r3 != r1
    assert hits() == 2
    both(q2, chains[:64])                # another query
    both(q, chains[:64])                 # the first again: its lists were overwritten
    both([q, q2] * 32, chains[:64], pairs=True)  # a multi-query call uses the same buffers
    assert hits() == 2
    both(q, chains[:64])
    assert hits() == 2                   # (a multi-query call never leaves a usable key behind)
    both(q, chains[:20])                 # the direct kernel (no lists), then the region correlate again
    both(q, chains[:64])
    both(q, chains[:130])                # a larger batch on the same window
    q.corrected_pose = p
    assert both(q, chains[:64]) == r1


@pytest.mark.parametrize("seed", [101, 102])
def test_long_lived_matcher_through_random_call_sequences(seed):
    """tests/soak.py: 70 random calls (single matches, one-query and multi-query batches of every size class, resident batches run
    again, pose writes in between) on ONE matcher give what a matcher that forgets everything between calls gives, bit for bit
    (scripts/dev/soak_calls.py runs the same for many seeds: 51 000 calls over 170 seeds without a difference at the end of round 5)."""
    from tests import soak
    assert soak.run(seed, 70, NCH=320, verbose=False) == 0
