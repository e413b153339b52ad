"""Development aid: one loop-closure query against n chains, the chains split over L matchers (a stream and a workspace each):
wall time per query of 16 pipelined queries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
query, chains = synth.loop_batch_scans(n)
for ch in chains:
    for s in ch:
        s.native(0)
for L in [int(v) for v in (sys.argv[2:] or ["1", "2", "4", "1", "4", "8"])]:
    ms = [ScanMatcher(None, loop=True) for _ in range(L)]
    per = (n + L - 1) // L
    bs = [m.make_batch(query, chains[i * per:(i + 1) * per]) for i, m in enumerate(ms)]
    def rounds(k):
        for r in range(k):
            for b in bs:
                b.run_async(False, False, slot=r)
        for m in ms:
            m.synchronize()
        for r in range(k):
            for b in bs:
                b.wait(r, per_chain=False)
    rounds(16); rounds(16)
    t = time.perf_counter()
    rounds(16)
    dt = (time.perf_counter() - t) / 16
    print("lanes %d: %.2f ms per query of %d chains" % (L, dt * 1e3, n))
    for m in ms:
        m.close()
