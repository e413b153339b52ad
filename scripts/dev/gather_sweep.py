"""Development aid: time the gather correlate of a 4096-item batch for several (jobs per wave, blocks per item, LDS budget)
settings -- debug options 15 / 17 / 20 -- on the cfg2 and the loop-closure lattice.  usage: gather_sweep.py [B] [loop]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.util import cfg2_scans
from tests.test_gpu_parity import _mk_native
from yag_slam_amd.scan_matching import ScanMatcher

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
loop = len(sys.argv) > 2 and sys.argv[2] == "loop"
q, base = cfg2_scans()
nq, nb = _mk_native(q), [_mk_native(b) for b in base]
chains = [nb for _ in range(B)]
pen, fine = (False, False) if loop else (True, True)
combos = [(0, 0, 0)] + [(na, parts, lds) for na in (1, 2, 3, 4) for parts in (1, 2, 3) for lds in (0,)]
if len(sys.argv) > 3:
    combos = [tuple(int(x) for x in c.split(",")) for c in sys.argv[3:]]
for na, parts, lds in combos:
    m = ScanMatcher(None, loop=loop)
    if na: m.debug_option(15, na)
    if parts: m.debug_option(17, parts)
    if lds: m.debug_option(20, lds)
    m.debug_option(14, 4)  # the gather correlate also on lattices the region correlate would take
    try:
        for _ in range(2):
            m.match_scan_batch(nq, chains, pen, fine)
        m.profile(True)
        for _ in range(4):
            m.match_scan_batch(nq, chains, pen, fine)
        ms, n = m.profile_read(0)
        cs, cn = m.profile_read(2)
        print("na %d parts %d lds %6d: correlate %8.1f us   call %8.1f us" % (na, parts, lds, ms / max(n, 1) * 1e3, cs / max(cn, 1) * 1e3), flush=True)
    except Exception as e:
        print("na %d parts %d lds %d: %s" % (na, parts, lds, e), flush=True)
    del m
