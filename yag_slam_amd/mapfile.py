"""yag-slam map files: `zlib(msgpack(dict))` with `___name`-tagged objects.

Reads and writes the on-disk format of `GraphSlam.to_file / from_file`
(/root/reference/yag_slam/graph_slam.py:77-130; object tagging /root/reference/yag_slam/serde.py:25-95;
scan fields /root/reference/yag_slam/models.py:41-53) so maps recorded with yag-slam can be replayed
through this matcher and maps built here open in yag-slam.

Layout of the unpacked dict:
    scans                       [ {ranges[], min_angle, max_angle, angle_increment, min_range, max_range,
                                   range_threshold, odom_pose: T, corrected_pose: T, num, ___name: "LocalizedRangeScan"} ]
    edges                       [ [from_num, to_num, {mean: T, covariance: 3x3, ___name: "LinkLabel"}] ]
    running_scans               [num]
    seq_matcher_config          {<config keys>, ___name: "ScanMatcherConfig"}
    loop_matcher_config         same or None
    scan_buffer_len, loop_search_dist, loop_search_min_chain_size, min_response_coarse, min_response_fine
with T = {x, y, z, qx, qy, qz, qw, ___name: "Transform"}; a planar pose is the quaternion
(0, 0, sin(yaw/2), cos(yaw/2)).
"""
import math
import zlib

import msgpack
import numpy as np

from .config import CONFIG_KEYS
from .mapping import LoopClosingMapper
from .models import LocalizedRangeScan
from .transform import Transform

TAG = "___name"
SCAN_FIELDS = ("min_angle", "max_angle", "angle_increment", "min_range", "max_range", "range_threshold")


def _pose_out(t):
    yaw = t.euler[-1]
    return {"x": t.x, "y": t.y, "z": getattr(t, "z", 0.0), "qx": 0.0, "qy": 0.0,
            "qz": math.sin(0.5 * yaw), "qw": math.cos(0.5 * yaw), TAG: "Transform"}


def _pose_in(d):
    # yaw of a general quaternion (ZYX convention); planar files have qx = qy = 0
    qx, qy, qz, qw = d.get("qx", 0.0), d.get("qy", 0.0), d["qz"], d["qw"]
    yaw = math.atan2(2.0 * (qw * qz + qx * qy), 1.0 - 2.0 * (qy * qy + qz * qz))
    return Transform(d["x"], d["y"], d.get("z", 0.0), yaw)


def _config_out(config):
    d = {k: getattr(config, k) for k in sorted(CONFIG_KEYS)}
    d[TAG] = "ScanMatcherConfig"
    return d


def _config_in(d):
    # the reference feeds every key except the tag back to the matcher constructor
    # (graph_slam.py:110-112); attributes of a newer karto_scanmatcher that this matcher does not
    # know are dropped rather than rejected
    return {k: v for k, v in d.items() if k in CONFIG_KEYS or k == "minimum_distance_penalty"}


def mapper_to_dict(mapper):
    scans = []
    for s in mapper.scans:
        d = {"ranges": np.asarray(s.ranges, dtype=np.float64).tolist()}
        for k in SCAN_FIELDS:
            d[k] = getattr(s, k)
        d["odom_pose"] = _pose_out(s.odom_pose)
        d["corrected_pose"] = _pose_out(s.corrected_pose)
        d["num"] = s.num
        d[TAG] = "LocalizedRangeScan"
        scans.append(d)
    edges = [[f, t, {"mean": _pose_out(mean), "covariance": np.asarray(cov, dtype=np.float64).tolist(), TAG: "LinkLabel"}]
             for f, t, mean, cov in mapper.constraints]
    return {
        "scans": scans,
        "edges": edges,
        "running_scans": [s.num for s in mapper.running_scans],
        "seq_matcher_config": _config_out(mapper.seq_matcher.config),
        "loop_matcher_config": _config_out(mapper.loop_matcher.config) if mapper.loop_matcher else None,
        "scan_buffer_len": mapper.scan_buffer_len,
        "loop_search_dist": mapper.loop_search_dist,
        "loop_search_min_chain_size": mapper.loop_search_min_chain_size,
        "min_response_coarse": mapper.min_response_coarse,
        "min_response_fine": mapper.min_response_fine,
    }


def _default_factory(config_dict, loop):
    from .scan_matching import ScanMatcher
    return ScanMatcher(config_dict, loop=loop)


def mapper_from_dict(d, matcher_factory=None, optimizer=None):
    """matcher_factory(config_dict, loop) -> matcher; default builds `ScanMatcher`s (needs the GPU)."""
    make = matcher_factory or _default_factory
    seq = make(_config_in(d["seq_matcher_config"]), False)
    loop = make(_config_in(d["loop_matcher_config"]), True) if d.get("loop_matcher_config") else None
    mp = LoopClosingMapper(seq, loop, d["scan_buffer_len"], d["loop_search_dist"], d["loop_search_min_chain_size"],
                           d["min_response_coarse"], d["min_response_fine"], optimizer=optimizer)
    for sd in d["scans"]:
        if sd.get(TAG) != "LocalizedRangeScan":
            raise ValueError("map file: scan entry tagged %r" % (sd.get(TAG),))
        s = LocalizedRangeScan(sd["ranges"], *[sd[k] for k in SCAN_FIELDS], 0.0, 0.0, 0.0)
        s.odom_pose = _pose_in(sd["odom_pose"])
        s.corrected_pose = _pose_in(sd["corrected_pose"])
        s.num = sd["num"]
        mp.add_vertex(s)
    for f, t, info in d["edges"]:
        # like graph_slam.py:118-124 the stored mean is kept (not recomputed from the poses)
        mean, cov = _pose_in(info["mean"]), info["covariance"]
        mp.adjacent[f].add(t)
        mp.adjacent[t].add(f)
        mp.constraints.append((f, t, mean, cov))
        if mp.opt is not None:
            mp.opt.add_constraint(f, t, mean.x, mean.y, mean.euler[-1], np.linalg.inv(np.array(cov)).tolist())
    mp.running_scans = [mp.scans[i] for i in d["running_scans"]]
    return mp


def dumps(mapper):
    return zlib.compress(msgpack.packb(mapper_to_dict(mapper)))


def loads(blob, matcher_factory=None, optimizer=None):
    return mapper_from_dict(msgpack.unpackb(zlib.decompress(blob)), matcher_factory, optimizer)


def to_file(mapper, path):
    with open(path, "wb") as f:
        f.write(dumps(mapper))


def from_file(path, matcher_factory=None, optimizer=None):
    with open(path, "rb") as f:
        return loads(f.read(), matcher_factory, optimizer)
