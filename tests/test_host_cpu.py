"""CPU-only checks: the C ABI library loads and exports every symbol include/yagmatch.h declares,
host-side logic (config surface, poses, scan model, sharding) behaves like the reference's."""
import math
import os
import re

import numpy as np
import pytest

from tests.util import REPO


def test_library_exports_every_declared_symbol():
    from yag_slam_amd import _capi
    L = _capi.lib()
    hdr = open(os.path.join(REPO, "include", "yagmatch.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ym_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(L, name), "libyagmatch.so lacks %s" % name
    assert declared == set(_capi.EXPORTS), declared ^ set(_capi.EXPORTS)
    assert L.ym_version() == 1


def test_struct_layouts_match_the_header():
    import ctypes as C
    from yag_slam_amd import _capi
    assert C.sizeof(_capi.YmConfig) == 11 * 8 + 2 * 4
    assert C.sizeof(_capi.YmScanDesc) == 8 + 8 + 6 * 8 + 3 * 8
    assert C.sizeof(_capi.YmResult) == 8 + 24 + 72 + 8 + 8 + 12 + 12 + 4 * 4
    assert C.sizeof(_capi.YmGridInfo) == 12 * 4 + 16


def test_no_device_fails_loudly():
    from yag_slam_amd import _capi
    from yag_slam_amd.scan_matching import ScanMatcher
    if _capi.lib().ym_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.YmError, match="no HIP device"):
        ScanMatcher()


def test_config_surface_matches_reference_defaults():
    from yag_slam_amd import config
    assert config.default_config["search_size"] == 0.5 and config.default_config["resolution"] == 0.01
    assert config.default_config_loop["search_size"] == 4.0 and config.default_config_loop["resolution"] == 0.05
    assert set(config.default_config) == set(config.CONFIG_KEYS) and len(config.CONFIG_KEYS) == 11
    c = config.make_config({"resolution": 0.02})
    assert c.resolution == 0.02 and c.smear_deviation == 0.05 and c.minimum_distance_penalty == 0.5
    assert config.make_config(c.as_dict()).as_dict() == c.as_dict()  # round-trips like serde.py:88-92
    with pytest.raises(AssertionError, match="Smear deviation"):
        config.make_config({"resolution": 0.01, "smear_deviation": 0.2})
    s = __import__("yag_slam_amd._capi", fromlist=["x"]).config_struct(c, "karto")
    assert s.resolution == 0.02 and s.use_response_expansion == 1 and s.semantics == 0


def test_transform_composes_like_an_odometry_prior():
    from yag_slam_amd.transform import Transform
    last_odom = Transform.from_position_euler(1.0, 2.0, 0, 0, 0, 0.3)
    new_odom = Transform.from_position_euler(1.5, 2.2, 0, 0, 0, 0.5)
    last_corr = Transform.from_position_euler(10.0, -3.0, 0, 0, 0, 1.3)
    prior = last_corr + (new_odom - last_odom)  # graph_slam.py:320-322
    # the relative motion is preserved in the corrected frame
    d = prior - last_corr
    e = new_odom - last_odom
    assert abs(d.x - e.x) < 1e-12 and abs(d.y - e.y) < 1e-12 and abs(d.euler[-1] - e.euler[-1]) < 1e-12
    assert abs(prior.euler[-1] - 1.5) < 1e-12
    ident = last_odom - last_odom
    assert abs(ident.x) < 1e-12 and abs(ident.y) < 1e-12 and abs(ident.euler[-1]) < 1e-12
    # tiny_tf's spelling: position + quaternion (what yag-slam's map files store).  A planar rotation round-trips; a
    # rotation out of the plane or a quaternion with components missing is refused, never read as some yaw
    t = Transform(1.0, 2.0, 0.0, 0.0, 0.0, math.sin(0.35), math.cos(0.35))
    assert abs(t.yaw - 0.7) < 1e-12
    t = Transform(1.0, 2.0, 0.0, qx=t.qx, qy=t.qy, qz=t.qz, qw=t.qw)
    assert abs(t.yaw - 0.7) < 1e-12
    with pytest.raises(ValueError):
        Transform(0.0, 0.0, 0.0, 0.1, 0.0, 0.0, 0.99)
    with pytest.raises(ValueError):
        Transform(0.0, 0.0, 0.0, qx=0.0, qy=0.3, qz=0.0, qw=0.95)
    with pytest.raises(TypeError):
        Transform(0.0, 0.0, 0.0, qz=0.5)


def test_scan_model_without_gpu():
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    z = np.load(os.path.join(REPO, "tests", "golden", "points_dirty.npz"))
    s = LocalizedRangeScan(z["ranges"], float(z["min_angle"]), 0.0, float(z["angle_increment"]), 0.05, 30.0,
                           float(z["range_threshold"]), *z["pose"])
    px, py = s.points()  # same rule as the reference's _get_point_readings (helpers.py:58-68)
    np.testing.assert_allclose(px, z["px"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(py, z["py"], rtol=0, atol=1e-12)
    s.corrected_pose = Transform.from_position_euler(1.0, 2.0, 0, 0, 0, 0.5)
    assert s.corrected_pose.x == 1.0 and s.odom_pose.x == float(z["pose"][0])
    c = s.copy()
    assert c.corrected_pose.euler[-1] == 0.5 and c.ranges is not s.ranges and c.num == 0
    s.num = 7
    assert s.num == 7
    lx, ly = s.points_local()
    assert len(lx) == len(px)
    # scans from JSON logs (models.py:110-116 of the reference): beams reversed unless told otherwise, threshold 0.9 x range_max
    d = {"ranges": [1.0, 2.0, 3.0, 40.0], "angle_min": -0.2, "angle_max": 0.1, "angle_increment": 0.1, "range_min": 0.05, "range_max": 30.0}
    j = LocalizedRangeScan.from_json(d, 1.0, 2.0, 0.5)
    assert list(j.ranges) == [40.0, 3.0, 2.0, 1.0] and j.range_threshold == 30.0 * 0.9 and j.min_angle == -0.2
    assert (j.corrected_pose.x, j.corrected_pose.y, j.corrected_pose.euler[-1]) == (1.0, 2.0, 0.5)
    assert list(LocalizedRangeScan.from_json(d, 0, 0, 0, invert=False).ranges) == [1.0, 2.0, 3.0, 40.0] and d["ranges"] == [1.0, 2.0, 3.0, 40.0]


def test_shard_ranges_cover_all_chains():
    from yag_slam_amd.dist import shard_range
    for n in (0, 1, 7, 8, 9, 4096, 4099):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pick_best_is_deterministic():
    from yag_slam_amd.dist import pick_best, RECORD
    r = np.zeros((4, RECORD))
    r[:, 0] = [0.5, 0.9, 0.9, 0.2]
    r[:, 1] = [0, 700, 300, 5]
    assert pick_best(r) == 2          # tie on response -> lowest chain id
    r[2, 1] = -1                      # empty shard never wins
    assert pick_best(r) == 1


def test_synthetic_scene_is_reproducible():
    from yag_slam_amd import synth
    a, b = synth.Scene(), synth.Scene()
    assert np.array_equal(a.segs, b.segs) and len(a.boxes) == 6
    r1 = a.scan_ranges((3.0, 3.0, 0.0), 3)
    r2 = b.scan_ranges((3.0, 3.0, 0.0), 3)
    assert np.array_equal(r1, r2) and r1.shape == (synth.N_BEAMS,)
    assert 0.5 < r1.min() and r1.max() < 10.0
    d = a.scan_ranges((3.0, 3.0, 0.0), 3, dirty=True)
    assert np.isnan(d).sum() == 10 and (d > synth.MAX_RANGE).sum() == 10
    truth, prior = synth.loop_trajectory(50)
    assert truth.shape == (50, 3) and np.all(np.hypot(*(truth[1:, :2] - truth[:-1, :2]).T) < 0.15)


def test_sequential_mapper_call_pattern():
    # graph_slam.py:306-339: odometry prior, penalty + fine, running chain of at most 10
    from collections import namedtuple
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    R = namedtuple("R", "best_pose response covariance meta")
    calls = []

    class Stub:
        def match_scan(self, q, base, pen, fine):
            calls.append((len(base), pen, fine, q.corrected_pose.x))
            return R(Transform(q.corrected_pose.x + 0.01, q.corrected_pose.y, 0, q.corrected_pose.euler[-1]), 1.0, None, {})

    mp = SequentialMapper(Stub())
    for i in range(14):
        s = LocalizedRangeScan([1.0] * 5, -1, 1, 0.5, 0, 10, 5, 0, 0, 0)
        s.odom_pose = Transform(0.1 * i, 0, 0, 0.0)
        mp.process_scan(s)
    assert len(calls) == 13 and all(c[1] and c[2] for c in calls)
    assert [c[0] for c in calls] == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10, 10]
    # prior = last corrected (which carries the accumulated +0.01 corrections) + odometry increment
    assert abs(calls[-1][3] - (0.1 * 13 + 0.01 * 12)) < 1e-9
    assert len(mp.running_scans) == 10 and mp.running_scans[-1].num == 13


def test_sequential_mapper_process_scans_falls_back_to_the_per_scan_loop():
    # a matcher plugin without map_sequence (the reference's own, the oracle): process_scans = process_scan in a loop
    from collections import namedtuple
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    R = namedtuple("R", "best_pose response covariance meta")

    class Stub:
        def match_scan(self, q, base, pen, fine):
            return R(Transform(q.corrected_pose.x + 0.01, q.corrected_pose.y, 0, q.corrected_pose.euler[-1]), 1.0, None, {"n": len(base)})

    def scans():
        out = []
        for i in range(14):
            s = LocalizedRangeScan([1.0] * 5, -1, 1, 0.5, 0, 10, 5, 0, 0, 0)
            s.odom_pose = Transform(0.1 * i, 0.02 * i, 0, 0.01 * i)
            out.append(s)
        return out
    a, b = SequentialMapper(Stub()), SequentialMapper(Stub())
    sa, sb = scans(), scans()
    ra = [a.process_scan(s) for s in sa]
    rb = b.process_scans(sb[:3]) + b.process_scans(sb[3:])
    assert rb[0] is None and [r.meta["n"] for r in rb[1:]] == [r.meta["n"] for r in ra[1:]]
    assert all((x.corrected_pose.x, x.corrected_pose.y, x.corrected_pose.euler[-1]) ==
               (y.corrected_pose.x, y.corrected_pose.y, y.corrected_pose.euler[-1]) for x, y in zip(sa, sb))
    assert [s.num for s in b.running_scans] == list(range(4, 14)) and len(b.results) == 13


def test_sequential_mapper_process_scans_keeps_what_was_matched_when_the_sequence_fails():
    # map_sequence raises at a scan Karto aborts on (or on a library error): the scans matched before it are done on both
    # sides, and the mapper's state is what the per-scan loop would have left when it raised there; scans that are not our
    # LocalizedRangeScan take the per-scan loop although the matcher has a map_sequence
    from collections import namedtuple
    import pytest
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    R = namedtuple("R", "best_pose response covariance meta")

    class Seq:
        def __init__(self, fail_at):
            self.fail_at, self.calls = fail_at, 0
        def match_scan(self, q, base, pen, fine):
            self.calls += 1
            return R(Transform(q.corrected_pose.x, q.corrected_pose.y, 0, q.corrected_pose.euler[-1]), 1.0, None, {"n": len(base)})
        def map_sequence(self, seq, start, buffer_len, pen, fine, device_chain):
            done = []
            for i in range(start, len(seq)):
                if i == self.fail_at:
                    self.sequence_done = done
                    raise RuntimeError("Mapper FATAL ERROR (scan %d)" % i)
                done.append(R(seq[i].corrected_pose, 1.0, None, {"i": i}))
            self.sequence_done = done
            return done

    def scans(n, cls=LocalizedRangeScan):
        out = []
        for i in range(n):
            s = cls([1.0] * 5, -1, 1, 0.5, 0, 10, 5, 0, 0, 0)
            s.odom_pose = Transform(0.1 * i, 0.0, 0, 0.0)
            out.append(s)
        return out
    m = SequentialMapper(Seq(fail_at=9), scan_buffer_len=4)
    ss = scans(14)
    m.process_scans(ss[:3])
    with pytest.raises(RuntimeError):
        m.process_scans(ss[3:])
    assert [r.meta["i"] for r in m.results[2:]] == [3, 4, 5, 6, 7, 8] and len(m.results) == 8
    assert [s.num for s in m.running_scans] == [5, 6, 7, 8]

    class Duck:  # (not our scan type: only the attributes process_scan touches)
        def __init__(self, i):
            self.odom_pose, self.corrected_pose, self.num = Transform(0.1 * i, 0.0, 0, 0.0), Transform(0, 0, 0, 0), 0
    m2 = SequentialMapper(Seq(fail_at=-1))
    out = m2.process_scans([Duck(i) for i in range(5)])
    assert out[0] is None and len(out) == 5 and m2.seq_matcher.calls == 4 and not hasattr(m2.seq_matcher, "sequence_done")


def _light_scan(num, pose):
    from yag_slam_amd.transform import Transform

    class S:
        pass
    s = S()
    s.num = num
    s.corrected_pose = Transform(pose[0], pose[1], 0.0, pose[2])
    return s


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_loop_chain_discovery_matches_reference(case):
    # fixtures come from the reference's own GraphSlam.find_possible_loop_closure_chains
    # (tests/golden/make_golden_chains.py); includes its squared-vs-plain distance comparison
    import os
    from yag_slam_amd.mapping import LoopClosingMapper
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loop_chains.npz"))
    poses, links = g[case + "_poses"], g[case + "_links"]
    dist, min_chain = g[case + "_params"]
    dist = int(dist) if float(dist).is_integer() else float(dist)
    mp = LoopClosingMapper(None, None, loop_search_dist=dist, loop_search_min_chain_size=int(min_chain))
    items = [_light_scan(i, p) for i, p in enumerate(poses)]
    eye = [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
    li = 0
    for i, it in enumerate(items):
        mp.add_vertex(it)
        if i > 0:
            assert tuple(links[li]) == (i - 1, i)
            mp.link_scans(items[i - 1], it, eye)
            li += 1
    for a, b in links[li:]:
        mp.link_scans(items[a], items[b], eye)
    mem, offs, qoffs = g[case + "_chain_members"], g[case + "_chain_offsets"], g[case + "_query_chain_offsets"]
    total = 0
    for qi, q in enumerate(g[case + "_queries"]):
        want = [list(mem[offs[c]:offs[c + 1]]) for c in range(qoffs[qi], qoffs[qi + 1])]
        got = [[s.num for s in ch] for ch in mp.find_possible_loop_closure_chains(items[q])]
        assert got == want, (case, q)
        total += len(want)
    assert total > 50


def test_loop_closing_mapper_two_stage_acceptance():
    # graph_slam.py:194-261: coarse (loop matcher, no penalty, no fine) gates on 0.35, then the fine
    # sequential match seeded with the coarse pose; first acceptable chain wins
    from collections import namedtuple
    from yag_slam_amd.mapping import LoopClosingMapper
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    R = namedtuple("R", "best_pose response covariance meta")
    eye = [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
    log = []

    class Seq:
        def match_scan(self, q, base, pen, fine):
            log.append(("seq", len(base), pen, fine, round(q.corrected_pose.x, 6)))
            p = q.corrected_pose
            return R(Transform(p.x, p.y, 0, p.euler[-1]), 0.9, eye, {})

    class Loop:
        def match_scan_batch(self, q, chains, pen, fine):
            log.append(("loop", len(chains), pen, fine))
            res = []
            for k, ch in enumerate(chains):
                good = k == len(chains) - 1          # only the last candidate chain passes the coarse gate
                res.append(R(Transform(ch[0].corrected_pose.x + 0.5, ch[0].corrected_pose.y, 0, 0.0), 0.8 if good else 0.1, eye, {}))
            return res, None

    mp = LoopClosingMapper(Seq(), Loop(), loop_search_dist=1.0, loop_search_min_chain_size=5)
    # out along +x for 4 m, jump back near the start along a parallel line 0.3 m away
    path = [(0.1 * i, 0.0) for i in range(40)] + [(3.9 - 0.1 * i, 0.3) for i in range(1, 40)]
    closed_at = []
    for i, (x, y) in enumerate(path):
        s = LocalizedRangeScan([1.0] * 5, -1, 1, 0.5, 0, 10, 5, 0, 0, 0)
        s.odom_pose = Transform(x, y, 0, 0.0)
        res, closed = mp.process_scan(s)
        if closed:
            closed_at.append(i)
    assert closed_at, "the return leg never closed a loop"
    first = closed_at[0]
    num, chain, rc, rf = mp.closures[0]
    assert num == first and rc.response == 0.8 and rf.response == 0.9
    # the fine stage ran on the sequential matcher without penalty, with refinement, from the coarse pose
    k = log.index(next(e for e in log if e[0] == "loop"))
    assert log[k][2:] == (False, False)
    assert log[k + 1][0] == "seq" and log[k + 1][2:4] == (False, True)
    assert abs(log[k + 1][4] - round(mp.scans[chain[0]].corrected_pose.x + 0.5, 6)) < 1e-6
    # a closure links the closing scan to the nearest member of the chain
    assert any(t == num and f in chain for f, t, _, _ in mp.constraints)
    # every scan got the odometry constraint to its predecessor
    assert all(any(f == n - 1 and t == n for f, t, _, _ in mp.constraints) for n in range(1, len(path)))


def _stub_factory(config_dict, loop):
    from yag_slam_amd.config import make_config

    class M:
        pass
    m = M()
    m.config = make_config(config_dict, loop=loop)
    m.loop = loop
    return m


def test_mapfile_reads_and_rewrites_reference_file():
    # tests/golden/graph_ref.bin was written by the reference's GraphSlam.binarize()
    # (tests/golden/make_golden_mapfile.py)
    import math, os, zlib, msgpack
    from yag_slam_amd import mapfile
    blob = open(os.path.join(os.path.dirname(__file__), "golden", "graph_ref.bin"), "rb").read()
    want = msgpack.unpackb(zlib.decompress(blob))
    mp = mapfile.loads(blob, _stub_factory)
    assert len(mp.scans) == 9 and [s.num for s in mp.running_scans] == [5, 6, 7, 8]
    assert (mp.scan_buffer_len, mp.loop_search_dist, mp.loop_search_min_chain_size) == (4, 2.5, 3)
    assert (mp.min_response_coarse, mp.min_response_fine) == (0.3, 0.5)
    assert mp.seq_matcher.config.search_size == 0.6 and mp.seq_matcher.config.smear_deviation == 0.03
    assert mp.loop_matcher.config.resolution == 0.05 and mp.loop_matcher.loop
    for s, sd in zip(mp.scans, want["scans"]):
        assert np.array_equal(np.asarray(s.ranges), np.asarray(sd["ranges"], dtype=np.float64), equal_nan=True)
        assert (s.min_angle, s.max_angle, s.angle_increment, s.range_threshold) == (
            sd["min_angle"], sd["max_angle"], sd["angle_increment"], sd["range_threshold"])
        for mine, theirs in ((s.odom_pose, sd["odom_pose"]), (s.corrected_pose, sd["corrected_pose"])):
            assert mine.x == theirs["x"] and mine.y == theirs["y"]
            assert abs(mine.euler[-1] - 2 * math.atan2(theirs["qz"], theirs["qw"])) < 1e-15
    assert math.isnan(mp.scans[3].ranges[5])
    assert sorted((f, t) for f, t, _, _ in mp.constraints) == sorted((e[0], e[1]) for e in want["edges"])
    assert 7 in mp.adjacent[1] and 1 in mp.adjacent[7]

    # writing it back gives the reference's dict (floats to the last bit except the quaternion
    # of a pose, which goes yaw -> quaternion once more)
    got = msgpack.unpackb(zlib.decompress(mapfile.dumps(mp)))
    assert got.keys() == want.keys()

    def same(a, b, path=""):
        if isinstance(a, dict):
            assert isinstance(b, dict) and a.keys() == b.keys(), path
            for k in a:
                same(a[k], b[k], path + "/" + str(k))
        elif isinstance(a, list):
            assert isinstance(b, list) and len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, path + "/" + str(i))
        elif isinstance(a, float) and math.isnan(a):
            assert math.isnan(b), path
        elif path.endswith(("/qz", "/qw")):
            assert abs(a - b) < 1e-15, path
        else:
            assert a == b, (path, a, b)
    same(got, want)


def test_mapfile_round_trip_of_a_mapper(tmp_path):
    from yag_slam_amd import mapfile
    from yag_slam_amd.mapping import LoopClosingMapper
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.transform import Transform
    mp = LoopClosingMapper(_stub_factory({}, False), None, scan_buffer_len=3)
    eye = [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
    prev = None
    for i in range(5):
        s = LocalizedRangeScan([1.0 + i] * 7, -1, 1, 0.3, 0, 10, 5, 0.2 * i, -0.1 * i, 0.3 * i - 3.0)
        s.num = i
        mp.add_vertex(s)
        if prev is not None:
            mp.link_scans(prev, s, eye)
        prev = s
    mp.running_scans = mp.scans[-3:]
    p = str(tmp_path / "m.bin")
    mapfile.to_file(mp, p)
    back = mapfile.from_file(p, _stub_factory)
    assert back.loop_matcher is None and len(back.scans) == 5
    assert [(f, t) for f, t, _, _ in back.constraints] == [(f, t) for f, t, _, _ in mp.constraints]
    for a, b in zip(mp.scans, back.scans):
        assert np.array_equal(a.ranges, b.ranges)
        assert abs(a.corrected_pose.euler[-1] - b.corrected_pose.euler[-1]) < 1e-15
        assert (a.corrected_pose.x, a.corrected_pose.y) == (b.corrected_pose.x, b.corrected_pose.y)


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under yag_slam_amd/ (Python or HIP) may import, link or call it
    import os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yag_slam_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|#include[^\n]*ym_oracle|orc_[a-z_]+\s*\(|libym_oracle|-lym_oracle)")
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert not pat.search(text), os.path.join(d, f)


def test_batch_results_is_a_lazy_sequence_over_the_records():
    """The per-chain results of a batch: built from the ym_result records when an entry is looked at, with the list
    behaviours callers of the reference's per-chain loop rely on (len, index, negative index, slice, iteration)."""
    import numpy as np
    from yag_slam_amd.scan_matching import BatchResults, _RESULT_DTYPE
    a = np.zeros(5, dtype=_RESULT_DTYPE)
    a["response"] = [0.1, 0.7, 0.7, 0.2, 0.0]
    a["pose"] = [[i, 2.0 * i, 0.1 * i] for i in range(5)]
    a["cov"] = [np.eye(3).ravel() * (i + 1) for i in range(5)]
    a["hypotheses"] = 14295
    a["coarse_dims"] = (26, 26, 21)
    a["fine_dims"] = (3, 3, 11)
    a["n_query_points"] = 1081
    res = BatchResults(a)
    assert len(res) == 5 and bool(res) and res.best() == 1  # (the first of equal responses, like ym_batch_wait)
    r = res[3]
    assert r.response == 0.2 and (r.best_pose.x, r.best_pose.y) == (3.0, 6.0) and abs(r.best_pose.euler[-1] - 0.3) < 1e-15
    assert r.covariance == [[4.0, 0.0, 0.0], [0.0, 4.0, 0.0], [0.0, 0.0, 4.0]]
    assert r.meta["coarse_dims"] == (26, 26, 21) and r.meta["fine_dims"] == (3, 3, 11) and r.meta["hypotheses"] == 14295
    assert res[-1].response == 0.0 and [x.response for x in res[1:3]] == [0.7, 0.7]
    assert [x.response for x in res] == [0.1, 0.7, 0.7, 0.2, 0.0]
    with pytest.raises(IndexError):
        res[5]
    assert not BatchResults(a[:0]) and BatchResults(a[:0]).best() == -1


def test_map_sequence_forgets_the_previous_calls_results_before_it_can_raise():
    # ScanMatcher.map_sequence publishes the scans it matched in `sequence_done`, which SequentialMapper.process_scans reads in
    # its except branch.  A call that raises BEFORE it reaches the library (a scan that is not resident, no odom_pose) must not
    # leave the previous call's list there: the mapper would commit those results again and take unmatched scans into its
    # running chain (round-4 advisor finding).
    import pytest
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.models import LocalizedRangeScan
    from yag_slam_amd.scan_matching import ScanMatcher
    from yag_slam_amd.transform import Transform
    m = ScanMatcher.__new__(ScanMatcher)  # (no device here: the call below fails before it needs one)
    m.device, m._m, m._lib = 0, None, None
    m.sequence_done = ["a result of the previous call"]
    with pytest.raises(TypeError):
        m.map_sequence([object(), object()], 1, 10)
    assert m.sequence_done == []
    # ... and the mapper on top of it: state unchanged by the failed call
    stale = ["stale-1", "stale-2"]

    class Seq:
        sequence_done = stale
        def match_scan(self, q, base, pen, fine):
            raise AssertionError("not reached")
        def map_sequence(self, seq, start, buffer_len, pen, fine, device_chain):
            return ScanMatcher.map_sequence(m, [object()] * len(seq), start, buffer_len, pen, fine, device_chain)

    def scan(i):
        s = LocalizedRangeScan([1.0] * 5, -1, 1, 0.5, 0, 10, 5, 0, 0, 0)
        s.odom_pose = Transform(0.1 * i, 0.0, 0, 0.0)
        return s
    seqm = Seq()
    seqm.sequence_done = m.sequence_done  # (what the real matcher's attribute would be)
    mp = SequentialMapper(seqm, scan_buffer_len=4)
    mp.running_scans = [scan(0), scan(1)]
    before = list(mp.running_scans)
    with pytest.raises(TypeError):
        mp.process_scans([scan(2), scan(3), scan(4)])
    assert mp.running_scans == before and mp.results == []


def test_oracle_under_sanitizers():
    # SURVEY section 5 proposes a sanitizer pass for the CPU side: the oracle built with -fsanitize=address,undefined
    # (`make -C oracle asan`) runs the reference's golden vectors (tests/test_oracle_golden.py) in a child process with the
    # sanitizer runtime preloaded; any report aborts the child (-fno-sanitize-recover, ASan's default abort on error).
    import os
    import shutil
    import subprocess
    import sys
    import pytest
    from tests.util import REPO
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               YM_ORACLE_LIB=os.path.join(REPO, "oracle", "_asan", "libym_oracle.so"))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_oracle_golden.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


def test_kernels_have_no_scratch_and_the_dominant_kernel_keeps_three_blocks_per_cu(tmp_path):
    """The product library, compiled here for gfx950 with -Rpass-analysis=kernel-resource-usage (hipcc cross-compiles without a GPU):
    no kernel may spill into scratch memory (a spill is 10 - 20 % of a latency-bound kernel: round 4 had four), and
    correlate_region_kernel<8, true> -- 60 % of the metric's step -- must stay within the 80 VGPRs that three blocks of eight waves
    per CU allow.  The lists' builder must keep the 64 VGPRs that let its block sit beside two region-correlate blocks."""
    import re
    import shutil
    import subprocess
    csrc = os.path.join(REPO, "yag_slam_amd", "csrc")
    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc on this host")
    out = str(tmp_path / "libyagmatch_check.so")
    p = subprocess.run(["make", "-s", "-C", csrc, "OUT=" + out, "FLAGS_EXTRA=-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    kernels = {}
    name = None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip() or m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            kernels[name][m.group(1).split(" ")[0]] = int(m.group(2))
    assert len(kernels) >= 60, len(kernels)
    spills = {k: v["ScratchSize"] for k, v in kernels.items() if v.get("ScratchSize", 0) > 0}
    assert not spills, "kernels with scratch: %s" % spills
    dom = [v for k, v in kernels.items() if k.startswith("void ym::correlate_region_kernel<8, true>")]
    assert len(dom) == 1 and dom[0]["VGPRs"] <= 80 and dom[0]["Occupancy"] >= 6, dom
    binp = [v for k, v in kernels.items() if k.startswith("void ym::bin_kernel<false, 18>")]
    assert len(binp) == 1 and binp[0]["VGPRs"] <= 64, binp
