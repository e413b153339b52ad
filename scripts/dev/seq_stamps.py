"""Development aid: the GPU timeline (phase stamps, 100 MHz) of ONE step of the sequential mapping loop (new query, one new
base scan in the chain), and the host split of ym_map_sequence (YM_DEBUG_HOST=1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.mapping import SequentialMapper
from yag_slam_amd.scan_matching import ScanMatcher
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
truth, scans = synth.trajectory_scans(N)
m = ScanMatcher()
for s in scans:
    s.native(0)
mp = SequentialMapper(m)
t = time.perf_counter()
mp.process_scans(scans[:N - 3 - 40])
dt = time.perf_counter() - t
print("library loop: %.1f us per step" % (dt * 1e6 / (N - 44)))
_, scans2 = synth.trajectory_scans(N)
for s in scans2:
    s.native(0)
mp2 = SequentialMapper(ScanMatcher())
t = time.perf_counter()
mp2.process_scans(scans2, device_chain=True)
dt = time.perf_counter() - t
print("device chain: %.1f us per step" % (dt * 1e6 / (N - 1)))
_, scans3 = synth.trajectory_scans(N)
for s in scans3:
    s.native(0)
m3 = ScanMatcher()
mp3 = SequentialMapper(m3)
mp3.process_scans(scans3[:200], device_chain=True)
m3.debug_stamps(True)
mp3.process_scans(scans3[200:300], device_chain=True)
st = m3.debug_stamps(False)
t0 = st[0]
nm = ["prep:start","prep:points","prep:trig","","rast:start","rast:scan","rast:rowpass","rast:end",
         "corr:start","corr:end","score:start","score:end","fine:start","fine:coarse","fine:cells","fine:end",
         "final:start","final:fties","prep:qend","final:end"]
print("chained step:", " ".join("%s %.2f" % (n_, (v - t0) / 100.0) for n_, v in zip(nm, st) if n_ and v), "idle before %.2f" % (st[28] / 100.0))
print("agreement with the library loop at scan %d:" % (N - 44), scans[N - 44].corrected_pose, scans2[N - 44].corrected_pose)
names = ["prep:start","prep:points","prep:trig","","rast:start","rast:scan","rast:rowpass","rast:end",
         "corr:start","corr:end","score:start","score:end","fine:start","fine:coarse","fine:cells","fine:end",
         "final:start","final:fties","prep:qend","final:end"]
m.debug_stamps(True)
mp.process_scans(scans[N - 3 - 40:N - 3 - 20])
st = m.debug_stamps(True)
print("library loop: device idle between final:end and the next prep:start %.2f us" % (st[28] / 100.0))
for s_ in scans[N - 3 - 20:N - 3]:
    mp.process_scan(s_)
st = m.debug_stamps(False)
print("python loop: device idle between final:end and the next prep:start %.2f us" % (st[28] / 100.0))
for k in range(3):
    m.debug_stamps(True)
    mp.process_scan(scans[N - 3 + k])
    st = m.debug_stamps(False)
    print("cache (hits, misses):", m.cache_stats())
    t0 = st[0]
    print(" ".join("%s %.2f" % (n, (v - t0) / 100.0) for n, v in zip(names, st) if n and v))
