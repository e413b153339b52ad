"""Matcher plugin: drop-in for /root/reference/yag_slam/scan_matching.py:29-42 (Scan2DMatcherCpp).

`ScanMatcher(config_dict, loop).match_scan(query, base_scans, penalty, do_fine)` returns the
reference's `ScanMatcherResult(response, covariance, best_pose, meta)` namedtuple; `.config` is the
attribute bag yag-slam serialises (graph_slam.py:82-83).  All numerics run in libyagmatch.so on
the MI355X; this file only marshals scans and results across the ctypes boundary.
"""
import ctypes as C
import os
import struct
from collections import namedtuple

import numpy as np

from . import _capi
from .config import default_config, default_config_loop, make_config
from .models import LocalizedRangeScan
from .transform import Transform

ScanMatcherResult = namedtuple('ScanMatcherResult', ['response', 'covariance', 'best_pose', 'meta'])


def _pose_of(scan):
    p = scan.corrected_pose
    return float(p.x), float(p.y), float(p.euler[-1])


def _desc_of(scan, keep):
    """ym_scan_desc from any duck-typed LocalizedRangeScan (host ranges are uploaded per call)."""
    d = _capi.YmScanDesc()
    r = np.ascontiguousarray(scan.ranges, dtype=np.float64)
    keep.append(r)
    d.ranges = r.ctypes.data_as(C.POINTER(C.c_double))
    d.n = int(r.shape[0])
    d.min_angle = float(scan.min_angle)
    d.max_angle = float(getattr(scan, "max_angle", scan.min_angle + (d.n - 1) * scan.angle_increment))
    d.angle_increment = float(scan.angle_increment)
    d.min_range = float(scan.min_range)
    d.max_range = float(getattr(scan, "max_range", 0.0))
    d.range_threshold = float(scan.range_threshold)
    d.pose[0], d.pose[1], d.pose[2] = _pose_of(scan)
    return d


# one unpack of the whole ym_result instead of ~30 ctypes field reads (a few microseconds of a 60 us match)
_RESULT_STRUCT = struct.Struct("<d3d9ddq3i3i4i")
assert _RESULT_STRUCT.size == C.sizeof(_capi.YmResult)


def _result(r):
    v = _RESULT_STRUCT.unpack_from(r)
    meta = {
        "coarse_response": v[13], "hypotheses": v[14], "coarse_dims": v[15:18], "fine_dims": v[18:21],
        "n_query_points": v[21], "expansions": v[22], "status": v[23],
    }
    return ScanMatcherResult(v[0], [list(v[4:7]), list(v[7:10]), list(v[10:13])], Transform(v[1], v[2], 0.0, v[3]), meta)


# numpy view of an array of ym_result (include/yagmatch.h): converting a batch result field by field through ctypes
# costs ~8 us per chain, more than the GPU spends on it
_RESULT_DTYPE = np.dtype([("response", "<f8"), ("pose", "<f8", (3,)), ("cov", "<f8", (9,)), ("coarse_response", "<f8"),
                          ("hypotheses", "<i8"), ("coarse_dims", "<i4", (3,)), ("fine_dims", "<i4", (3,)),
                          ("n_query_points", "<i4"), ("expansions", "<i4"), ("status", "<i4"), ("reserved", "<i4")])
assert _RESULT_DTYPE.itemsize == C.sizeof(_capi.YmResult)


class BatchResults(object):
    """The per-chain results of a batch as a read-only sequence of ScanMatcherResult, built when an entry is looked at:
    a loop closure asks for the few chains above its thresholds, and 4096 result objects cost more Python time (4.5 ms)
    than the GPU spends on the 4096 matches.  `.array` is the numpy record view of the ym_result array
    (include/yagmatch.h): response, pose, cov, coarse_response, hypotheses, coarse_dims, fine_dims, n_query_points,
    expansions, status -- the form to filter and sort in."""

    __slots__ = ("array",)

    def __init__(self, array):
        self.array = array

    def __len__(self):
        return int(self.array.shape[0])

    def _one(self, i):
        r = self.array[i]
        p = r["pose"]
        return ScanMatcherResult(float(r["response"]), r["cov"].reshape(3, 3).tolist(), Transform(float(p[0]), float(p[1]), 0.0, float(p[2])),
                                 {"coarse_response": float(r["coarse_response"]), "hypotheses": int(r["hypotheses"]),
                                  "coarse_dims": tuple(r["coarse_dims"].tolist()), "fine_dims": tuple(r["fine_dims"].tolist()),
                                  "n_query_points": int(r["n_query_points"]), "expansions": int(r["expansions"]),
                                  "status": int(r["status"])})

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._one(j) for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("chain index out of range")
        return self._one(i)

    def __iter__(self):
        return (self._one(i) for i in range(len(self)))

    def best(self):
        """index of the chain with the highest response (the first of equals, like ym_batch_wait's best_chain)"""
        return int(np.argmax(self.array["response"])) if len(self) else -1


def _results(per, check=False):
    """BatchResults over a copy of a ctypes array of YmResult.  check: raise like match_scan does when any chain reports
    Karto's "unable to find best position / index out of range"."""
    a = np.frombuffer(per, dtype=_RESULT_DTYPE).copy()
    if check and a["status"].any():
        bad = int(np.flatnonzero(a["status"])[0])
        raise _capi.YmError(int(a["status"][bad]), "Mapper FATAL ERROR - unable to find best position / index out of range "
                            "(chain %d)" % bad)
    return BatchResults(a)


class ScanMatcher(object):
    """MI355X correlative scan matcher behind yag-slam's matcher plugin surface."""

    def __init__(self, config_dict=None, loop=False, semantics="karto", device=0):
        cfg = default_config if not loop else default_config_loop
        cfg = cfg.copy()
        if config_dict:
            cfg.update(config_dict)
        self.config = make_config(cfg)
        self.semantics = semantics
        self.device = int(device)
        self._lib = _capi.lib()
        self._cfg = _capi.config_struct(self.config, semantics)
        self._m = self._lib.ym_create(C.byref(self._cfg), self.device)
        if not self._m:
            raise _capi.YmError(-1, _capi.last_error())
        # development only, and only on request: YM_DEVELOPMENT=1 YM_DEBUG_OPTIONS="32=1,33=2" -> ym_debug_option on every
        # matcher of the process.  Without YM_DEVELOPMENT an inherited YM_DEBUG_OPTIONS changes nothing in a library user.
        if os.environ.get("YM_DEVELOPMENT") == "1":
            for kv in filter(None, (t.strip() for t in os.environ.get("YM_DEBUG_OPTIONS", "").split(","))):
                try:
                    k, v = (int(t) for t in kv.split("="))
                except ValueError:
                    raise ValueError("YM_DEBUG_OPTIONS: cannot read %r (expected option=value, both integers)" % kv)
                self.debug_option(k, v)

    def close(self):
        if getattr(self, "_m", None):
            self._lib.ym_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- handles -------------------------------------------------------------------------
    def _native(self, scan):
        """ym_scan* of a resident scan (our LocalizedRangeScan), else None.  Like the reference's C++ twin
        (/root/reference/yag_slam/models.py:67-75) the device twin follows the pose through the `corrected_pose`
        setter, so nothing is re-sent here."""
        if isinstance(scan, LocalizedRangeScan):
            return scan.native(self.device)
        return None

    def _push_pose(self, scan):
        """re-send the current corrected pose (for poses mutated in place, behind the setter's back)"""
        h = self._native(scan)
        if h is not None:
            x, y, t = _pose_of(scan)
            _capi.check(self._lib.ym_scan_set_pose(h, x, y, t))

    # ---- the plugin call -------------------------------------------------------------------
    def match_scan(self, query, base_scans, penalty=True, do_fine=False):
        res = _capi.YmResult()
        dev = self.device
        handles = [s._native if (type(s) is LocalizedRangeScan and s._native is not None and s._native_device == dev)
                   else self._native(s) for s in (query, *base_scans)]
        if None not in handles:
            nb = len(handles) - 1
            arr = (C.c_void_p * max(1, nb))(*handles[1:])
            rc = self._lib.ym_match_scans(self._m, handles[0], arr, nb, 1 if penalty else 0, 1 if do_fine else 0, C.byref(res))
            if rc:
                _capi.check(rc)
        else:
            keep = []
            q = _desc_of(query, keep)
            b = (_capi.YmScanDesc * max(1, len(base_scans)))(*[_desc_of(s, keep) for s in base_scans])
            _capi.check(self._lib.ym_match(self._m, C.byref(q), b, len(base_scans), int(bool(penalty)),
                                           int(bool(do_fine)), C.byref(res)))
        if res.status != 0:
            raise _capi.YmError(res.status, "Mapper FATAL ERROR - unable to find best position / index out of range")
        return _result(res)

    def map_sequence(self, scans, start, buffer_len, penalty=True, do_fine=True, device_chain=False):
        """The matcher calls of `GraphSlam.process_scan` (/root/reference/yag_slam/graph_slam.py:320-337) for
        scans[start:], scans[:start] being the running chain so far: odometry prior from `odom_pose`, match against the
        last `buffer_len` scans, `corrected_pose` = the match's pose -- one library call (`ym_match_scans` in a host
        loop without Python).  Every scan must be resident (our LocalizedRangeScan).  Returns the results of
        scans[start:]; raises like `match_scan` at the first scan Karto would abort on (the scans before it are done).
        device_chain: no host round trip between the steps either (the device hands each step's pose to the next;
        include/yagmatch.h) -- the priors are then composed with the device's cos / sin, so poses agree with the
        step-by-step form to rounding, not bit for bit."""
        self.sequence_done = []  # (before anything can raise: a caller's except branch must not see the previous call's results)
        n = len(scans)
        handles = (C.c_void_p * max(1, n))(*[self._require_native(s) for s in scans])
        odom = np.empty((max(1, n), 3), dtype=np.float64)
        for i, s in enumerate(scans):
            p = s.odom_pose
            odom[i, 0], odom[i, 1], odom[i, 2] = p.x, p.y, p.euler[-1]
        per = (_capi.YmResult * max(1, n))()
        done = C.c_int32(0)
        rc = self._lib.ym_map_sequence(self._m, handles, odom.ctypes.data_as(C.POINTER(C.c_double)), n, int(start),
                                       int(buffer_len), int(bool(penalty)), int(bool(do_fine)), int(bool(device_chain)),
                                       per, C.byref(done))
        first = max(int(start), 1)
        res = _results(per)[first:done.value] if done.value > first else []
        for s, r in zip(scans[first:done.value], res):
            s._corrected_pose = r.best_pose  # (the device twin already has it)
        self.sequence_done = res  # (the scans matched before an error: SequentialMapper.process_scans commits them)
        if rc:  # (a library error: the scans matched so far keep their results, Python and device twins agree)
            _capi.check(rc)
        if done.value < n:
            bad = scans[done.value]
            pose = (C.c_double * 3)()
            _capi.check(self._lib.ym_scan_get_pose(bad._native, pose))
            bad._corrected_pose = Transform(pose[0], pose[1], 0.0, pose[2])  # the prior, where the per-scan path leaves it
            raise _capi.YmError(per[done.value].status, "Mapper FATAL ERROR - unable to find best position / index out of "
                                "range (scan %d of the sequence)" % done.value)
        return res

    def process_scan(self, query, chain, penalty=True, do_fine=True):
        """`GraphSlam.process_scan`'s matcher work for one scan in one library call (`ym_process_scan`): the odometry
        prior from `chain[-1]` and the two `odom_pose`s, the match, `query.corrected_pose` = the result.  Resident scans
        only (returns None otherwise: the caller does it the long way).  Same bits as the three separate steps."""
        dev = self.device
        scans = (query, *chain)
        for s in scans:
            if type(s) is not LocalizedRangeScan or s._native is None or s._native_device != dev:
                if not isinstance(s, LocalizedRangeScan):
                    return None
                s.native(dev)
        n = len(chain)
        arr = (C.c_void_p * n)(*[s._native for s in chain])
        lo, qo = chain[-1].odom_pose, query.odom_pose
        a = (C.c_double * 3)(lo.x, lo.y, lo.euler[-1])
        b = (C.c_double * 3)(qo.x, qo.y, qo.euler[-1])
        res = _capi.YmResult()
        rc = self._lib.ym_process_scan(self._m, query._native, arr, n, a, b, 1 if penalty else 0, 1 if do_fine else 0, C.byref(res))
        if rc:
            _capi.check(rc)
        pose = (C.c_double * 3)()
        if res.status != 0:
            _capi.check(self._lib.ym_scan_get_pose(query._native, pose))
            query._corrected_pose = Transform(pose[0], pose[1], 0.0, pose[2])  # (the prior, where the long way leaves it)
            raise _capi.YmError(res.status, "Mapper FATAL ERROR - unable to find best position / index out of range")
        r = _result(res)
        query._corrected_pose = r.best_pose  # (the device twin already has it)
        return r

    def sequence_stats(self):
        """(device-chained segments, segments cut short by a fault, synchronous steps) of map_sequence so far"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _capi.check(self._lib.ym_sequence_stats(self._m, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def match_scan_batch(self, query, chains, penalty=False, do_fine=False):
        """One query against many candidate chains (the loop of graph_slam.py:217-236 in one call).
        Returns (per_chain_results, best_index)."""
        flat, offs = [], [0]
        for ch in chains:
            flat.extend(ch)
            offs.append(len(flat))
        hq = self._require_native(query)
        hs = (C.c_void_p * max(1, len(flat)))(*self._handles(flat))
        co = (C.c_int32 * len(offs))(*offs)
        per = (_capi.YmResult * len(chains))()
        best = _capi.YmResult()
        bi = C.c_int32(-1)
        _capi.check(self._lib.ym_match_batch(self._m, hq, hs, co, len(chains), int(bool(penalty)), int(bool(do_fine)),
                                             per, C.byref(best), C.byref(bi)))
        return _results(per, check=True), int(bi.value)

    def make_batch(self, query, chains):
        """Reusable (query, chains) batch for the pipelined loop-closure path."""
        return MatchBatch(self, query, chains)

    def match_pairs(self, queries, chains, penalty=True, do_fine=False):
        """len(queries) INDEPENDENT matches in one enqueue: item i = `match_scan(queries[i], chains[i], penalty, do_fine)`
        (N x /root/reference/yag_slam/graph_slam.py:326 -- N robots, or N segments of a log replayed side by side), item for
        item bit-identical to the single calls.  Returns the per-item results (BatchResults)."""
        if len(queries) != len(chains):
            raise ValueError("match_pairs needs one chain per query (%d queries, %d chains)" % (len(queries), len(chains)))
        if not queries:
            return BatchResults(np.zeros(0, dtype=_RESULT_DTYPE))
        flat, offs = [], [0]
        for ch in chains:
            flat.extend(ch)
            offs.append(len(flat))
        hq = (C.c_void_p * len(queries))(*self._handles(queries))
        hs = (C.c_void_p * max(1, len(flat)))(*self._handles(flat))
        co = (C.c_int32 * len(offs))(*offs)
        per = (_capi.YmResult * len(chains))()
        _capi.check(self._lib.ym_match_pairs(self._m, hq, hs, co, len(chains), int(bool(penalty)), int(bool(do_fine)), per))
        return _results(per, check=True)

    def make_pairs_batch(self, queries, chains):
        """Reusable form of `match_pairs` (resident on the device, runnable many times; poses are read at every run)."""
        return MatchBatch(self, queries, chains, pairs=True)

    def _handles(self, scans):
        """ym_scan* of every scan; the twins that already live on this device are taken without a call"""
        dev, out = self.device, []
        for s in scans:
            h = getattr(s, "_native", None)
            if h is None or getattr(s, "_native_device", None) != dev or not isinstance(s, LocalizedRangeScan):
                h = self._require_native(s)
            out.append(h)
        return out

    def _require_native(self, scan):
        h = self._native(scan)
        if h is None:
            raise TypeError("match_scan_batch needs yag_slam_amd.models.LocalizedRangeScan instances")
        return h

    # ---- one match split over several matchers by coarse angle (yag_slam_amd.dist.AngleSplitMatcher) -----------------
    def coarse_dims(self):
        """(nx, ny, ntheta) of the coarse lattice of this matcher's config"""
        d = (C.c_int32 * 3)()
        _capi.check(self._lib.ym_coarse_dims(self._m, d))
        return tuple(d[:])

    def slice_begin(self, query, base_scans, penalty, do_fine, k_begin, k_end, dev_resp, dev_probs):
        """Enqueue the match up to the score stage for coarse angles [k_begin, k_end); responses and per-(x, y) maxima go
        to the caller's device buffers (raw pointers).  No host wait."""
        hq = self._require_native(query)
        arr = (C.c_void_p * max(1, len(base_scans)))(*self._handles(base_scans))
        _capi.check(self._lib.ym_match_slice_begin(self._m, hq, arr, len(base_scans), int(bool(penalty)), int(bool(do_fine)),
                                                   int(k_begin), int(k_end), C.c_void_p(dev_resp), C.c_void_p(dev_probs)))

    def slice_finish(self):
        """The rest of the match on the completed volume; returns the ScanMatcherResult of the whole match."""
        res = _capi.YmResult()
        _capi.check(self._lib.ym_match_slice_finish(self._m, C.byref(res)))
        if res.status != 0:
            raise _capi.YmError(res.status, "Mapper FATAL ERROR - unable to find best position / index out of range")
        return _result(res)

    # ---- match against a prebuilt map (Scan2DMatcherPy.match_scan_sets_with_map, scan_matching.py:124-173) ------------
    def correlation_grid_from_occupancy(self, map_im, occupied_value=0):
        """Device counterpart of `occupancy_grid_map_to_correlation_grid(map_im, res, smear_deviation, occupied_value)`
        (/root/reference/yag_slam/helpers.py:24-34) with this matcher's resolution and smear: returns a resident
        CorrelationMap (`.to_numpy()` gives the float grid the reference function returns)."""
        im = np.ascontiguousarray(map_im, dtype=np.uint8)
        if im.ndim != 2:
            raise ValueError("occupancy image must be 2-D")
        h = self._lib.ym_map_from_occupancy(self._m, im.ctypes.data_as(C.POINTER(C.c_uint8)), im.shape[1], im.shape[0],
                                            im.strides[0], int(occupied_value))
        if not h:
            raise _capi.YmError(-1, _capi.last_error())
        return CorrelationMap(self, h)

    def upload_correlation_grid(self, cgrid):
        """A correlation grid computed elsewhere (2-D float array, 0..1) as a resident CorrelationMap."""
        g = np.ascontiguousarray(cgrid, dtype=np.float64)
        if g.ndim != 2:
            raise ValueError("correlation grid must be 2-D")
        h = self._lib.ym_map_from_grid(self._m, g.ctypes.data_as(C.POINTER(C.c_double)), g.shape[1], g.shape[0])
        if not h:
            raise _capi.YmError(-1, _capi.last_error())
        return CorrelationMap(self, h)

    def match_scan_sets_with_map(self, cgrid, ox, oy, query_scans, penalty=True, do_fine=True, coarse=None):
        """The reference's signature (scan_matching.py:124): the point readings of all `query_scans` are matched as one
        set against the map `cgrid` (CorrelationMap or 2-D float array) whose cell (0, 0) lies at world (ox, oy).
        Returns ScanMatcherResult(response, covariance, [corrected pose of every query scan], meta).
        `coarse`: dict overriding the coarse pass the reference hard-codes (xy_search 0.25, xy_step 0.01,
        angle_search 0.1, angle_step 0.01, grid_resolution 0.05, penalize False)."""
        own = not isinstance(cgrid, CorrelationMap)
        mp = self.upload_correlation_grid(cgrid) if own else cgrid
        try:
            hs = (C.c_void_p * len(query_scans))(*[self._require_native(q) for q in query_scans])
            cs = None
            if coarse:
                cs = _capi.YmMapSearch(float(coarse.get("xy_search", 0.25)), float(coarse.get("xy_step", 0.01)),
                                       float(coarse.get("angle_search", 0.1)), float(coarse.get("angle_step", 0.01)),
                                       float(coarse.get("grid_resolution", 0.05)), int(bool(coarse.get("penalize", False))), 0)
            res = _capi.YmResult()
            _capi.check(self._lib.ym_match_map(self._m, mp._h, float(ox), float(oy), hs, len(query_scans), int(bool(penalty)),
                                               int(bool(do_fine)), C.byref(cs) if cs else None, C.byref(res)))
        finally:
            if own:
                mp.close()
        if res.status != 0:
            raise _capi.YmError(res.status, "empty search lattice or no query point")
        r = _result(res)
        # scan_matching.py:136-139,167-173: the search centre is the mean query position with heading 0; every query is
        # moved by (corrected centre - centre)
        xs = [float(q.corrected_pose.x) for q in query_scans]
        ys = [float(q.corrected_pose.y) for q in query_scans]
        oxy = Transform.from_position_euler(sum(xs) / float(len(xs)), sum(ys) / float(len(ys)), 0, 0, 0, 0)
        diff = r.best_pose - oxy
        r.meta["centre"] = (oxy.x, oxy.y, 0.0)
        r.meta["corrected_centre"] = (r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1])
        return ScanMatcherResult(r.response, r.covariance, [q.corrected_pose + diff for q in query_scans], r.meta)

    # ---- pipelined form ----------------------------------------------------------------------
    def match_scan_async(self, query, base_scans, penalty=True, do_fine=False, slot=0):
        hq = self._require_native(query)
        arr = (C.c_void_p * max(1, len(base_scans)))(*self._handles(base_scans))
        _capi.check(self._lib.ym_match_scans_async(self._m, hq, arr, len(base_scans), int(bool(penalty)),
                                                   int(bool(do_fine)), int(slot)))

    def wait(self, slot=0):
        res = _capi.YmResult()
        _capi.check(self._lib.ym_wait(self._m, int(slot), C.byref(res)))
        return _result(res)

    def synchronize(self):
        _capi.check(self._lib.ym_synchronize(self._m))

    def set_stream(self, stream_handle):
        """Run on a caller-owned HIP stream (e.g. torch.cuda.current_stream().cuda_stream); None = own."""
        _capi.check(self._lib.ym_set_stream(self._m, C.c_void_p(stream_handle) if stream_handle else None))

    # ---- introspection for the parity tests --------------------------------------------------
    def debug_grid(self, item=0):
        info = _capi.YmGridInfo()
        _capi.check(self._lib.ym_debug_grid_info(self._m, item, C.byref(info)))
        buf = np.zeros((info.height, info.pitch), dtype=np.uint8)
        _capi.check(self._lib.ym_debug_grid(self._m, item, buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size))
        return buf[:, :info.width], info

    def debug_sums(self, pass_=0, item=0, dims=None):
        nx, ny, nt = dims
        buf = np.zeros((nt, ny, nx), dtype=np.uint32)
        _capi.check(self._lib.ym_debug_sums(self._m, item, pass_, buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.size))
        return buf

    def debug_query_local(self, item=0, cap=8192):
        buf = np.zeros((cap, 2))
        n = C.c_int32()
        _capi.check(self._lib.ym_debug_query_local(self._m, item, buf.ctypes.data_as(C.POINTER(C.c_double)), cap,
                                                   C.byref(n)))
        return buf[:n.value].copy()

    def debug_cells(self, item=0, cap=1 << 16):
        """Window cells (wx, wy) of the base readings that were rasterised, [slot][beam]; INT32_MIN = filtered."""
        buf = np.zeros((cap, 2), dtype=np.int32)
        max_n = C.c_int32()
        _capi.check(self._lib.ym_debug_cells(self._m, item, buf.ctypes.data_as(C.POINTER(C.c_int32)), buf.size,
                                             C.byref(max_n)))
        return buf.reshape(-1, 2), max_n.value

    def debug_option(self, option, value):
        _capi.check(self._lib.ym_debug_option(self._m, int(option), int(value)))

    def debug_stamps(self, enable=True):
        buf = (C.c_uint64 * 32)()
        _capi.check(self._lib.ym_debug_stamps(self._m, int(bool(enable)), buf, 32))
        return list(buf)

    def cache_stats(self):
        """(hits, misses) of the matcher's point cache since it was created"""
        h, ms = C.c_int64(0), C.c_int64(0)
        _capi.check(self._lib.ym_cache_stats(self._m, C.byref(h), C.byref(ms)))
        return int(h.value), int(ms.value)

    def debug_counters(self):
        """dict of the matcher's counters since it was created (include/yagmatch.h, ym_debug_counters)"""
        buf = (C.c_int64 * 8)()
        _capi.check(self._lib.ym_debug_counters(self._m, buf, 8))
        return dict(yag_fast_items=int(buf[0]), yag_fallback_items=int(buf[1]), yag_pairs_checked=int(buf[2]),
                    yag_pairs_failed=int(buf[3]), list_cache_hits=int(buf[4]),
                    last_correlate={-1: None, 0: "correlate_kernel", 1: "correlate_region_kernel", 2: "gather_kernel"}[int(buf[5])])

    def profile(self, on=True):
        _capi.check(self._lib.ym_profile_enable(self._m, int(bool(on))))

    def profile_read(self, which=0, reset=True):
        ms, n = C.c_double(), C.c_int64()
        _capi.check(self._lib.ym_profile_read(self._m, which, C.byref(ms), C.byref(n), int(bool(reset))))
        return ms.value, n.value


class CorrelationMap(object):
    """A correlation grid resident on the device (ym_map): built from an occupancy image or uploaded."""

    def __init__(self, matcher, handle):
        self.m, self._h = matcher, handle
        w, h = C.c_int32(), C.c_int32()
        _capi.check(matcher._lib.ym_map_size(handle, C.byref(w), C.byref(h)))
        self.shape = (h.value, w.value)

    def to_numpy(self):
        out = np.empty(self.shape, dtype=np.float64)
        _capi.check(self.m._lib.ym_map_read(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.m._lib.ym_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchBatch(object):
    """One query against many candidate chains, resident on the device, runnable many times.

    Replaces the serial `for chain in chains: loop_matcher.match_scan(scan, chain, False, False)` of
    /root/reference/yag_slam/graph_slam.py:217-220 by one enqueue per batch."""

    def __init__(self, matcher, query, chains, pairs=False):
        """pairs: `query` is a sequence of queries, item i = query[i] against chains[i] (ScanMatcher.make_pairs_batch)"""
        self.m = matcher
        self.query = query
        self.chains = [list(c) for c in chains]
        self._queries = list(query) if pairs else [query]
        if pairs and len(self._queries) != len(self.chains):
            raise ValueError("a pairs batch needs one chain per query")
        flat, offs = [], [0]
        for ch in self.chains:
            flat.extend(ch)
            offs.append(len(flat))
        self._flat = flat
        hs = (C.c_void_p * max(1, len(flat)))(*matcher._handles(flat))
        co = (C.c_int32 * len(offs))(*offs)
        if pairs:
            hq = (C.c_void_p * max(1, len(self._queries)))(*matcher._handles(self._queries))
            self._h = matcher._lib.ym_pairs_create(matcher._m, hq, hs, co, len(self.chains))
        else:
            hq = matcher._require_native(query)
            self._h = matcher._lib.ym_batch_create(matcher._m, hq, hs, co, len(self.chains))
        if not self._h:
            raise _capi.YmError(-1, _capi.last_error())
        self.n = len(self.chains)

    @classmethod
    def from_handles(cls, matcher, query_handles, scan_handles, offsets):
        """A pairs batch over raw ym_scan* arrays (numpy uint64; models.ScanBlock.handles): item i = query_handles[i] against
        scan_handles[offsets[i]:offsets[i + 1]].  No Python object per scan; the caller keeps the scans alive while the batch is used."""
        self = cls.__new__(cls)
        self.m, self.query, self.chains, self._queries, self._flat = matcher, None, None, None, None
        qh = np.ascontiguousarray(query_handles, dtype=np.uint64)
        sh = np.ascontiguousarray(scan_handles, dtype=np.uint64)
        co = np.ascontiguousarray(offsets, dtype=np.int32)
        if co.shape[0] != qh.shape[0] + 1:
            raise ValueError("a pairs batch needs one chain per query")
        self._h = matcher._lib.ym_pairs_create(matcher._m, qh.ctypes.data_as(C.POINTER(C.c_void_p)), sh.ctypes.data_as(C.POINTER(C.c_void_p)),
                                               co.ctypes.data_as(C.POINTER(C.c_int32)), int(qh.shape[0]))
        if not self._h:
            raise _capi.YmError(-1, _capi.last_error())
        self.n = int(qh.shape[0])
        return self

    def close(self):
        if getattr(self, "_h", None):
            try:
                self.m._lib.ym_batch_destroy(self._h)
            except Exception:
                pass
            self._h = None

    def __del__(self):
        self.close()

    def push_poses(self):
        """Write the scans' current corrected poses through to their device twins."""
        scans = self._queries + self._flat
        hs = self.m._handles(scans)
        xyz = np.array([_pose_of(s) for s in scans], dtype=np.float64).reshape(len(scans), 3)
        _capi.check(self.m._lib.ym_scans_set_poses((C.c_void_p * len(hs))(*hs), xyz.ctypes.data_as(C.POINTER(C.c_double)), len(hs)))

    def run_async(self, penalty=False, do_fine=False, slot=0, chain_id_base=0, dev_best_out=None):
        _capi.check(self.m._lib.ym_batch_run_async(self.m._m, self._h, int(bool(penalty)), int(bool(do_fine)), int(slot),
                                                   int(chain_id_base), C.c_void_p(dev_best_out) if dev_best_out else None))

    def wait(self, slot=0, per_chain=True):
        per = (_capi.YmResult * self.n)() if per_chain else None
        best = _capi.YmResult()
        bi = C.c_int32(-1)
        _capi.check(self.m._lib.ym_batch_wait(self.m._m, int(slot), per, C.byref(best), C.byref(bi)))
        return (_results(per, check=True) if per_chain else None), _result(best), int(bi.value)


# names the reference exports (scan_matching.py:32,224)
Scan2DMatcherCpp = ScanMatcher
Scan2DMatcher = ScanMatcher
