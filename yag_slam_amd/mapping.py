"""Sequential mapping driver: the caller side of the hot path.

A minimal restatement of the call pattern of `GraphSlam.process_scan`
(/root/reference/yag_slam/graph_slam.py:306-339): odometry prior = last corrected pose composed
with the odometry increment, `seq_matcher.match_scan(query, running_scans, True, True)`, running
chain of the last `scan_buffer_len` scans (default 10, graph_slam.py:47,336-337).  No pose graph, no
optimiser, no loop closure here -- this only drives the matcher the way yag-slam does, with every
scan resident on the device (one upload per new scan, poses written through).
"""
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
        odom_diff = query.odom_pose - last_scan.odom_pose
        query.corrected_pose = last_scan.corrected_pose + odom_diff
        res = self.seq_matcher.match_scan(query, self.running_scans, True, True)
        query.corrected_pose = res.best_pose
        self.running_scans.append(query)
        self.running_scans = self.running_scans[-self.scan_buffer_len:]
        self.results.append(res)
        return res
