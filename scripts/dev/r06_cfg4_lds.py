"""Round 6, the one attempt on the gather correlate (cfg4): does staging larger regions -- up to a whole class image per work item --
buy anything?  The region size follows the LDS a block may use (debug option 20); the kernel's registers allow two blocks of eight
waves per CU whatever the LDS, so up to 80 KB per block cost no occupancy.  Prints the plan (YM_DEBUG_PLAN) and the kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.models import native_many
from yag_slam_amd.scan_matching import ScanMatcher
n = 4096
query, chains = synth.loop_batch_scans(n)
native_many([query] + [s for ch in chains for s in ch], 0)
for lds in (0, 65000, 80000, 100000):
    m = ScanMatcher(None, loop=True)
    if lds:
        m.debug_option(20, lds)
    b = m.make_batch(query, chains)
    for r in range(4):
        b.run_async(False, False, slot=r)
    for r in range(4):
        b.wait(r, per_chain=False)
    m.profile(True)
    for r in range(8):
        b.run_async(False, False, slot=r)
    for r in range(8):
        b.wait(r, per_chain=False)
    ms, k = m.profile_read(0)
    cms, ck = m.profile_read(2)
    print("LDS per block %6d: gather correlate %.1f us, call %.1f us (over %d)" % (lds, ms / max(k, 1) * 1e3, cms / max(ck, 1) * 1e3, k), flush=True)
    m.close()
