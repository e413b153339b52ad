import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yag_slam_amd import synth
from yag_slam_amd.scan_matching import ScanMatcher
query, chains = synth.loop_batch_scans(4096)
m = ScanMatcher(None, loop=True)
for ch in chains:
    for s in ch:
        s.native(0)
b = m.make_batch(query, chains)
m.profile(True)
for i in range(3):
    t = time.perf_counter()
    b.run_async(False, False, slot=0)
    t1 = time.perf_counter()
    import ctypes as C
    from yag_slam_amd import _capi
    from yag_slam_amd.scan_matching import _results, _result
    perc = (_capi.YmResult * b.n)()
    bestc = _capi.YmResult()
    bic = C.c_int32(-1)
    ta = time.perf_counter()
    _capi.check(m._lib.ym_batch_wait(m._m, 0, perc, C.byref(bestc), C.byref(bic)))
    tb = time.perf_counter()
    per = _results(perc, check=True)
    tc = time.perf_counter()
    print("  ym_batch_wait %.2f ms, _results %.2f ms" % ((tb - ta) * 1e3, (tc - tb) * 1e3))
    t2 = time.perf_counter()
    print("GPU ms: correlate %.2f raster %.2f call %.2f" % tuple(m.profile_read(w)[0] for w in range(3)))
    print("run_async %.2f ms, wait incl. results %.2f ms, expansions max %d, count %d" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, int(per.array["expansions"].max()), int((per.array["expansions"] > 0).sum())))
