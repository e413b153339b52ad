"""Development aid: a device-chained sequence alone (for kernel traces)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.mapping import SequentialMapper
from yag_slam_amd.scan_matching import ScanMatcher
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
chain = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
truth, scans = synth.trajectory_scans(N)
m = ScanMatcher()
for o_, v_ in zip(sys.argv[3::2], sys.argv[4::2]):  # debug options: pairs after the two arguments
    m.debug_option(int(o_), int(v_))
for s in scans:
    s.native(0)
mp = SequentialMapper(m)
t = time.perf_counter()
mp.process_scans(scans, device_chain=chain)
dt = time.perf_counter() - t
print("%s: %.1f us per step" % ("device chain" if chain else "library loop", dt * 1e6 / (N - 1)))
