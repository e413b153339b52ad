"""Run on the GPU box after `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl -o ovl -- python3 bench.py --only cfg2x
--no-production-legs --only-headline --steps 8 --warmup 2`: how the kernels of the four lanes share the device inside the timed steps --
for the last 60 % of the ym:: kernel activity: wall time, sum of kernel durations, time with 0 / 1 / 2 / 3+ kernels in flight, and per
kernel its total share.  Prints a small markdown table (kept as profiles/r05_step_overlap.md)."""
import csv, glob, collections, sys
f = glob.glob("gpurun_out/ovl/**/ovl_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "ym::" not in n or "structure_kernel" in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", ""), r.get("Queue_Id", "?")))
rows.sort()
# the timed steps = the busy segment (no gap of 3 ms without a kernel in flight) with the most region-correlate launches: the self-checks,
# the warm-up and the one-enqueue-at-a-time pass after the timed region are separated from it by drains
segs, cur, end = [], [], None
for r in rows:
    if end is not None and r[0] > end + 3000000:
        segs.append(cur); cur = []
    cur.append(r)
    end = r[1] if end is None else max(end, r[1])
segs.append(cur)
sel = max(segs, key=lambda sg: sum("correlate_region" in r[2] for r in sg))
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(depth, 4)] += t - last
    last = t
    depth += d
wall = ev[-1][0] - ev[0][0]
busy = sum(e - s for s, e, n, q in sel)
per = collections.Counter()
cnt = collections.Counter()
for s, e, n, q in sel:
    per[n] += e - s; cnt[n] += 1
queues = sorted({q for s, e, n, q in sel})
print("| | |\n|---|---|")
print("| window | %.1f ms of the timed steps, %d kernel launches on %d hardware queues |" % (wall * 1e-6, len(sel), len(queues)))
print("| sum of kernel durations / wall | %.2f |" % (busy / wall))
for d in range(5):
    print("| time with %s kernels in flight | %.1f %% |" % (("%d" % d) if d < 4 else "4 or more", 100.0 * hist[d] / wall))
print("\n| kernel | launches | share of the summed kernel time | mean duration while the lanes overlap, us |\n|---|---|---|---|")
for n, v in per.most_common(8):
    print("| `%s` | %d | %.1f %% | %.0f |" % (n, cnt[n], 100.0 * v / busy, v / cnt[n] * 1e-3))
