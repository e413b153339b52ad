"""Sequential mapping driver: the caller side of the hot path.

A minimal restatement of the call pattern of `GraphSlam.process_scan`
(/root/reference/yag_slam/graph_slam.py:306-339): odometry prior = last corrected pose composed
with the odometry increment, `seq_matcher.match_scan(query, running_scans, True, True)`, running
chain of the last `scan_buffer_len` scans (default 10, graph_slam.py:47,336-337).  No pose graph, no
optimiser, no loop closure here -- this only drives the matcher the way yag-slam does, with every
scan resident on the device (one upload per new scan, poses written through).
"""
from .models import LocalizedRangeScan, set_corrected_poses
from .transform import Transform


class SequentialMapper(object):
    def __init__(self, seq_matcher, scan_buffer_len=10):
        self.seq_matcher = seq_matcher
        self.scan_buffer_len = scan_buffer_len
        self.running_scans = []
        self.results = []

    def process_scan(self, scan):
        """scan: LocalizedRangeScan with .odom_pose set.  Returns the matcher result (None for the first scan)."""
        query = scan
        if len(self.running_scans) == 0:
            query.num = 0
            self.running_scans.append(query)
            return None
        last_scan = self.running_scans[-1]
        query.num = last_scan.num + 1
        fast = getattr(self.seq_matcher, "process_scan", None)
        res = fast(query, self.running_scans, True, True) if fast is not None else None
        if res is None:  # (any matcher plugin: the three steps of graph_slam.py:320-327)
            odom_diff = query.odom_pose - last_scan.odom_pose
            query.corrected_pose = last_scan.corrected_pose + odom_diff
            res = self.seq_matcher.match_scan(query, self.running_scans, True, True)
            query.corrected_pose = res.best_pose
        self.running_scans.append(query)
        self.running_scans = self.running_scans[-self.scan_buffer_len:]
        self.results.append(res)
        return res


    def process_scans(self, scans, device_chain=False):
        """`process_scan` for every scan of a trajectory with the loop itself inside the library (`ym_map_sequence`: no
        Python between two matches).  Same poses, results and running chain as calling `process_scan` scan by scan;
        returns the list of results (None for the very first scan of a map).  device_chain: the device also hands each
        step's pose to the next without a host round trip (ScanMatcher.map_sequence); poses then agree to rounding."""
        scans = list(scans)
        out = []
        if scans and not self.running_scans:
            out.append(self.process_scan(scans.pop(0)))
        if not scans:
            return out
        native = getattr(self.seq_matcher, "map_sequence", None)
        resident = all(isinstance(s, LocalizedRangeScan) for s in self.running_scans + scans)
        if native is None or not resident:  # (any matcher plugin, any scan type: scan by scan, as process_scan does)
            return out + [self.process_scan(s) for s in scans]
        start = len(self.running_scans)
        seq = self.running_scans + scans
        for k in range(start, len(seq)):
            seq[k].num = seq[k - 1].num + 1
        try:
            res = native(seq, start, self.scan_buffer_len, True, True, device_chain)
        except Exception:
            # the scans matched before the failing one are done (poses final on both sides): the mapper's state is what
            # the per-scan loop would have left when it raised at that scan
            res = list(getattr(self.seq_matcher, "sequence_done", None) or [])
            self.running_scans = (seq[:start + len(res)])[-self.scan_buffer_len:]
            self.results.extend(res)
            raise
        self.running_scans = seq[-self.scan_buffer_len:]
        self.results.extend(res)
        return out + res


def _dist2(a, b):
    """squared planar distance between two scans' corrected poses (helpers.py:383-386)"""
    pa, pb = a.corrected_pose, b.corrected_pose
    return (pa.x - pb.x) ** 2 + (pa.y - pb.y) ** 2


class PoseBuckets(object):
    """Coarse spatial index of scans by corrected pose.

    Same candidate set as the reference's `RadiusHashSearch.crude_radius_search`
    (/root/reference/yag_slam/helpers.py:396-431): scans are bucketed by `(int(x/res), int(y/res))`
    (truncation toward zero) and a query returns every scan whose bucket CORNER `(ix*res, iy*res)` lies
    within `radius + res` of the query position.  Only the bounded block of buckets that can satisfy
    that test is visited instead of the whole table."""

    def __init__(self, res):
        self.res = float(res)
        self.buckets = {}

    def key(self, pose):
        return (int(pose.x / self.res), int(pose.y / self.res))

    def add(self, scan):
        self.buckets.setdefault(self.key(scan.corrected_pose), []).append(scan)

    def rebuild(self, scans):
        self.buckets = {}
        for s in scans:
            self.add(s)

    def near(self, pose, radius):
        reach = radius + self.res
        r2 = reach * reach
        lo_x, hi_x = int((pose.x - reach) / self.res) - 1, int((pose.x + reach) / self.res) + 1
        lo_y, hi_y = int((pose.y - reach) / self.res) - 1, int((pose.y + reach) / self.res) + 1
        out = []
        for ix in range(lo_x, hi_x + 1):
            for iy in range(lo_y, hi_y + 1):
                got = self.buckets.get((ix, iy))
                if got and (ix * self.res - pose.x) ** 2 + (iy * self.res - pose.y) ** 2 < r2:
                    out.extend(got)
        return out


class LoopClosingMapper(SequentialMapper):
    """`GraphSlam.process_scan` + `try_to_close_loop` as the matcher sees them
    (/root/reference/yag_slam/graph_slam.py:194-261,272-339), with the per-chain coarse loop of
    `try_to_close_loop` (graph_slam.py:217-220) issued as ONE batched enqueue.

    What is kept: the odometry prior, the running chain, the constraint bookkeeping (previous scan,
    closest scan of the running chain, closest scan of the closing chain), candidate-chain discovery
    with the reference's exact rules (including its comparison of a SQUARED distance with
    `loop_search_dist`, graph_slam.py:290, and its dropping of the last candidate of the sorted list,
    :284), the two-stage acceptance (coarse loop matcher without penalty >= min_response_coarse, then
    the fine sequential matcher seeded with the coarse pose), first acceptable chain wins.
    What is not here: the sparse pose adjustment the reference runs after a closure (`SPA2d`, third
    party, out of the hot path).  An `optimizer` with the same four calls (`add_node`,
    `add_constraint`, `compute`, `nodes`) may be plugged in; without one the closure only corrects
    the closing scan.

    The reference rejects a low FINE response only when `verbose` is set (`... and self.verbose`,
    graph_slam.py:240); `reject_low_fine=None` follows that, True/False overrides it."""

    def __init__(self, seq_matcher, loop_matcher, scan_buffer_len=10, loop_search_dist=3,
                 loop_search_min_chain_size=10, min_response_coarse=0.35, min_response_fine=0.45,
                 verbose=False, optimizer=None, reject_low_fine=None):
        super().__init__(seq_matcher, scan_buffer_len)
        self.loop_matcher = loop_matcher
        self.loop_search_dist = loop_search_dist
        self.loop_search_min_chain_size = loop_search_min_chain_size
        self.min_response_coarse = min_response_coarse
        self.min_response_fine = min_response_fine
        self.verbose = verbose
        self.reject_low_fine = verbose if reject_low_fine is None else reject_low_fine
        self.opt = optimizer
        self.scans = []           # vertex list, index == scan.num
        self.adjacent = []        # adjacency sets by scan.num
        self.constraints = []     # (from_num, to_num, mean Transform, covariance)
        self.index = PoseBuckets(loop_search_dist)
        self.closures = []        # (scan.num, chain nums, coarse result, fine result)

    # ---- graph bookkeeping (graph_slam.py:132-192) ----
    def add_vertex(self, scan):
        assert scan.num == len(self.scans)
        self.scans.append(scan)
        self.adjacent.append(set())
        self.index.add(scan)
        if self.opt is not None:
            p = scan.corrected_pose
            self.opt.add_node(p.x, p.y, p.euler[-1], scan.num)

    def link_scans(self, from_scan, to_scan, covariance):
        if to_scan.num in self.adjacent[from_scan.num] or from_scan.num == to_scan.num:
            return False
        mean = to_scan.corrected_pose - from_scan.corrected_pose
        self.adjacent[from_scan.num].add(to_scan.num)
        self.adjacent[to_scan.num].add(from_scan.num)
        self.constraints.append((from_scan.num, to_scan.num, mean, covariance))
        if self.opt is not None:
            import numpy as np
            self.opt.add_constraint(from_scan.num, to_scan.num, mean.x, mean.y, mean.euler[-1],
                                    np.linalg.inv(np.array(covariance)).tolist())
        return True

    def link_to_closest_scan_in_chain(self, scan, chain, covariance):
        self.link_scans(min(chain, key=lambda s: _dist2(s, scan)), scan, covariance)

    def near_linked(self, scan):
        """scans reachable from `scan` through constraints without leaving the loop_search_dist disc
        (`do_breadth_first_traversal` with `make_near_scan_visitor`, graph.py:73-101, graph_slam.py:32-39)"""
        limit = self.loop_search_dist ** 2
        seen, todo, inside = {scan.num}, [scan.num], set()
        while todo:
            n = todo.pop()
            if not _dist2(self.scans[n], scan) < limit:
                continue
            inside.add(n)
            for a in self.adjacent[n]:
                if a not in seen:
                    seen.add(a)
                    todo.append(a)
        return inside

    def find_possible_loop_closure_chains(self, scan):
        """graph_slam.py:274-304"""
        excluded = self.near_linked(scan)
        cand = sorted(self.index.near(scan.corrected_pose, self.loop_search_dist), key=lambda s: s.num)
        chains, current = [], []
        for s, nxt in zip(cand, cand[1:]):
            if s.num == scan.num or s.num in excluded:
                current = []
                continue
            if _dist2(scan, s) <= self.loop_search_dist:
                current.append(s)
            if len(current) >= self.loop_search_min_chain_size:
                chains.append(current)
                current = []
            if nxt.num - s.num > 1:
                current = []
        if current:
            chains.append(current)
        return chains

    def try_to_close_loop(self, scan):
        """graph_slam.py:194-261"""
        if not self.loop_matcher:
            return False
        chains = self.find_possible_loop_closure_chains(scan)
        if not chains:
            return False
        if hasattr(self.loop_matcher, "match_scan_batch"):
            coarse, _ = self.loop_matcher.match_scan_batch(scan, chains, False, False)
        else:
            coarse = [self.loop_matcher.match_scan(scan, chain, False, False) for chain in chains]
        for chain, res_coarse in zip(chains, coarse):
            if res_coarse.response < self.min_response_coarse:
                continue
            tmp = scan.copy()
            tmp.corrected_pose = res_coarse.best_pose
            res = self.seq_matcher.match_scan(tmp, chain, False, True)
            if res.response < self.min_response_fine and self.reject_low_fine:
                continue
            scan.corrected_pose = res.best_pose
            self.link_to_closest_scan_in_chain(scan, chain, res.covariance)
            self.closures.append((scan.num, [s.num for s in chain], res_coarse, res))
            self.run_opt()
            return True
        return False

    def make_occupancy_grid(self, resolution=0.05, range_threshold=12):
        """graph_slam.py:341-342: every vertex scan ray-traced into one grid (on the device)"""
        from .occupancy import create_occupancy_grid
        return create_occupancy_grid(self.scans, resolution, range_threshold, device=getattr(self.seq_matcher, "device", 0))

    def run_opt(self):
        """graph_slam.py:262-272; a no-op without an optimizer except for the spatial index refresh"""
        if self.opt is not None:
            self.opt.compute(100, 1.0e-4, True, 1.0e-9, 50)
            n = min(len(self.opt.nodes), len(self.scans))  # (zip's length, as graph_slam.py:268)
            set_corrected_poses(self.scans[:n], [Transform(node.x, node.y, 0.0, node.yaw) for node in self.opt.nodes[:n]])
        self.index.rebuild(self.scans)

    def process_scan(self, scan):
        """Returns (result, closed) like graph_slam.py:306-339; (None, None) for the first scan."""
        query = scan
        if len(self.running_scans) == 0:
            query.num = 0
            self.running_scans.append(query)
            self.add_vertex(query)
            return None, None
        last_scan = self.running_scans[-1]
        query.num = last_scan.num + 1
        query.corrected_pose = last_scan.corrected_pose + (query.odom_pose - last_scan.odom_pose)
        res = self.seq_matcher.match_scan(query, self.running_scans, True, True)
        query.corrected_pose = res.best_pose
        self.add_vertex(query)
        self.link_scans(last_scan, query, res.covariance)
        if self.loop_matcher:
            self.link_to_closest_scan_in_chain(query, self.running_scans, res.covariance)
        closed = self.try_to_close_loop(query)
        self.running_scans.append(query)
        self.running_scans = self.running_scans[-self.scan_buffer_len:]
        self.results.append(res)
        return res, closed
