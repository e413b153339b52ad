"""Development aid: how many scans carry a trusted trigger-chain structure (bench trajectories)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth, _capi
L = _capi.lib()
_, scans = synth.trajectory_scans(2000)
q, chains = synth.loop_batch_scans(512)
allscans = scans + [s for ch in chains for s in ch]
n = [0, 0]
for s in allscans:
    h = s.native(0)
    for sem in (0, 1):
        n[sem] += L.ym_scan_structure_trusted(h, sem)
print("scans %d: trusted karto %d (%.2f %%), yagpy %d (%.2f %%)" % (len(allscans), n[0], 100.0 * n[0] / len(allscans), n[1], 100.0 * n[1] / len(allscans)))
