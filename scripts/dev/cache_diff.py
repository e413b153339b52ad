import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher
from yag_slam_amd.transform import Transform
q, base = cfg2_scans()
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
extra = [_mk_native(b) for b in base]
m, ref = ScanMatcher(), ScanMatcher()
ref.debug_option(7, 1)
plans = [
    ("single", nb), ("single", nb), ("move", 3), ("single", nb), ("single", nb[2:7]),
    ("batch", [nb, nb[:5], nb[5:], nb]), ("move", 0), ("move", 9), ("batch", [nb, nb[:5], nb[5:], nb]),
    ("batch", [nb[:4] + extra[:3], extra, nb]), ("move", 5), ("single", nb[::-1]),
    ("batch", [nb] * 9 + [extra] * 3), ("batch", [nb] * 9 + [extra] * 3),
]
k = 0
for step, (kind, arg) in enumerate(plans):
    if kind == "move":
        p = nb[arg].corrected_pose
        k += 1
        nb[arg].corrected_pose = Transform(p.x + 0.013 * k, p.y - 0.007 * k, 0.0, p.euler[-1] + 0.011 * k)
        continue
    if kind == "single":
        m.match_scan(nq, arg, True, True), ref.match_scan(nq, arg, True, True)
        chains = [arg]
    else:
        m.match_scan_batch(nq, arg, True, True), ref.match_scan_batch(nq, arg, True, True)
        chains = arg
    for item, ch in enumerate(chains):
        ca, mn = m.debug_cells(item)
        cb, _ = ref.debug_cells(item)
        mb = max(len(c) for c in chains)
        ca = ca[:mb * mn].reshape(mb, mn, 2); cb = cb[:mb * mn].reshape(mb, mn, 2)
        for s in range(mb):
            if not np.array_equal(ca[s], cb[s]):
                d = np.argwhere((ca[s] != cb[s]).any(axis=1)).ravel()
                print("step", step, kind, "item", item, "slot", s, "used" if s < len(ch) else "UNUSED", "ndiff", len(d), d[:5], ca[s][d[:3]].tolist(), cb[s][d[:3]].tolist())
print("done")
