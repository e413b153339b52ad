"""Development aid: how many 32-beam groups of the staged correlate kernel fit their LDS rectangle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher
q, base = cfg2_scans()
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
m = ScanMatcher()
m.debug_option(0, 1)
m.match_scan(nq, nb, True, True)
m.debug_stamps(True)
m.match_scan(nq, nb, True, True)
st = m.debug_stamps(False)
g, l, by = st[29], st[30], st[31]
print("groups %d, staged %d (%.1f %%), staged bytes per staged group %.0f (gathered 32 x 676 = 21632)" % (g, l, 100.0 * l / max(g, 1), by / max(l, 1)))
