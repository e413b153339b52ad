"""Development aid: library-loop step time with debug options (pairs: option value ...)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.mapping import SequentialMapper
from yag_slam_amd.scan_matching import ScanMatcher
N = 1000
truth, scans = synth.trajectory_scans(N)
for s in scans:
    s.native(0)
args = [int(v) for v in sys.argv[1:]]
for rep in range(3):
    m = ScanMatcher()
    for o, v in zip(args[0::2], args[1::2]):
        m.debug_option(o, v)
    truth, scans = synth.trajectory_scans(N)
    for s in scans:
        s.native(0)
    mp = SequentialMapper(m)
    t = time.perf_counter()
    mp.process_scans(scans)
    dt = time.perf_counter() - t
    print("options %s: library loop %.1f us per step" % (args, dt * 1e6 / (N - 1)))
