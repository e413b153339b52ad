#!/usr/bin/env python3
"""Golden vectors for loop-closure candidate-chain discovery, produced by the reference's own
`GraphSlam.find_possible_loop_closure_chains` (/root/reference/yag_slam/graph_slam.py:274-304).

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_chains.py

The reference's graph_slam.py, graph.py and helpers.py are imported unmodified.  Third-party modules
this image lacks are replaced by throw-away stand-ins created in a temp directory at run time:
`numba`, `karto_scanmatcher`, `cv2`, `msgpack` is real, `sba_cpp.SPA2d` (records nothing), and
`tiny_tf.tf.Transform` (x, y, euler and a planar `-`).  The graph is filled through the reference's
own `add_vertex` / `link_scans`, so vertex order, adjacency, the spatial hash and the chain rules
are all the reference's code.  Output: tests/golden/loop_chains.npz (poses, links, queries and the
chains the reference returned) -- data only.
"""
import math
import os
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


def _install_stubs():
    d = tempfile.mkdtemp(prefix="ymgold_stubs_")
    for m in ("numba", "karto_scanmatcher", "tiny_tf", "cv2", "sba_cpp"):
        os.makedirs(os.path.join(d, m))
    with open(os.path.join(d, "numba", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            def njit(*a, **k):
                if len(a) == 1 and callable(a[0]) and not k:
                    return a[0]
                return lambda fn: fn
            prange = range
        """))
    with open(os.path.join(d, "karto_scanmatcher", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            class ScanMatcherConfig: pass
            class Wrapper: pass
            class LaserScanConfig: pass
            class LocalizedRangeScan: pass
            class Pose2:
                def __init__(self, x=0.0, y=0.0, yaw=0.0):
                    self.x, self.y, self.yaw = x, y, yaw
            def create_occupancy_grid(*a, **k): raise NotImplementedError
        """))
    with open(os.path.join(d, "sba_cpp", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            class SPA2d:
                def __init__(self): self.nodes = []
                def add_node(self, *a): pass
                def add_constraint(self, *a): pass
                def compute(self, *a): pass
        """))
    with open(os.path.join(d, "tiny_tf", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "tiny_tf", "tf.py"), "w") as f:
        f.write(textwrap.dedent("""
            import math
            class Transform:
                def __init__(self, x, y, yaw):
                    self.x, self.y, self.euler = x, y, (0.0, 0.0, yaw)
                @classmethod
                def from_position_euler(cls, x, y, z, r, p, yaw):
                    return cls(x, y, yaw)
                def __sub__(self, o):
                    c, s = math.cos(o.euler[-1]), math.sin(o.euler[-1])
                    dx, dy = self.x - o.x, self.y - o.y
                    return Transform(c * dx + s * dy, -s * dx + c * dy, self.euler[-1] - o.euler[-1])
        """))
    with open(os.path.join(d, "cv2", "__init__.py"), "w") as f:
        f.write("")
    sys.path.insert(0, REF)
    sys.path.insert(0, d)


_install_stubs()
from yag_slam.graph_slam import GraphSlam  # noqa: E402  (the reference)
from tiny_tf.tf import Transform  # noqa: E402  (stand-in)


class Item:
    def __init__(self, num, pose):
        self.num = num
        self.corrected_pose = Transform(*pose)


def trajectory(seed, n, rx, ry, step, jitter):
    rng = np.random.default_rng(seed)
    per = 2 * math.pi * math.sqrt((rx * rx + ry * ry) / 2.0)
    ph = 2 * math.pi * np.arange(n) * step / per
    x = rx * np.cos(ph) + rng.normal(0, jitter, n)
    y = ry * np.sin(ph) + rng.normal(0, jitter, n)
    # cross the origin so that int() truncation toward zero in the spatial hash is exercised
    return np.stack([x - 0.7, y + 0.4, np.zeros(n)], axis=1)


def main():
    out = {}
    cases = [
        dict(name="a", seed=1, n=260, rx=5.0, ry=3.0, step=0.2, jitter=0.02, dist=3, min_chain=10, extra_links=[]),
        dict(name="b", seed=2, n=300, rx=2.3, ry=1.3, step=0.1, jitter=0.01, dist=1.0, min_chain=10, extra_links=[(5, 120)]),
        dict(name="c", seed=3, n=200, rx=4.0, ry=0.6, step=0.15, jitter=0.03, dist=2, min_chain=4, extra_links=[(10, 150), (20, 90)]),
    ]
    for c in cases:
        poses = trajectory(c["seed"], c["n"], c["rx"], c["ry"], c["step"], c["jitter"])
        g = GraphSlam(object(), object(), loop_search_dist=c["dist"], loop_search_min_chain_size=c["min_chain"])
        items = [Item(i, p) for i, p in enumerate(poses)]
        links = []
        eye = [[1.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]]
        for i, it in enumerate(items):
            g.add_vertex(it)
            if i > 0:
                g.link_scans(items[i - 1], it, it.corrected_pose, eye)
                links.append((i - 1, i))
        for a, b in c["extra_links"]:
            g.link_scans(items[a], items[b], items[b].corrected_pose, eye)
            links.append((a, b))
        queries = list(range(0, c["n"], 7)) + [c["n"] - 1]
        flat, offs, qoffs = [], [0], [0]
        for q in queries:
            chains = g.find_possible_loop_closure_chains(items[q])
            for ch in chains:
                flat.extend(s.num for s in ch)
                offs.append(len(flat))
            qoffs.append(len(offs) - 1)
        n_chains = len(offs) - 1
        print("case %s: %d queries, %d chains, %d members" % (c["name"], len(queries), n_chains, len(flat)))
        k = c["name"]
        out[k + "_poses"] = poses
        out[k + "_links"] = np.array(links, dtype=np.int32)
        out[k + "_params"] = np.array([c["dist"], c["min_chain"]], dtype=np.float64)
        out[k + "_queries"] = np.array(queries, dtype=np.int32)
        out[k + "_chain_members"] = np.array(flat, dtype=np.int32)
        out[k + "_chain_offsets"] = np.array(offs, dtype=np.int32)
        out[k + "_query_chain_offsets"] = np.array(qoffs, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "loop_chains.npz"), **out)


if __name__ == "__main__":
    main()
