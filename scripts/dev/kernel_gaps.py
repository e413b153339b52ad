"""Development aid: idle time between consecutive kernels of a rocprofv3 kernel trace CSV (one stream assumed)."""
import csv, sys, statistics
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
gaps = {}
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = (s1 - e0) / 1e3
    if 0 <= g < 200:
        gaps.setdefault((n0[-28:], n1[-28:]), []).append(g)
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 20:
        print("%-30s -> %-30s n=%4d median %6.1f us total %8.1f us" % (k[0], k[1], len(v), statistics.median(v), sum(v)))
        tot += sum(v)
print("total of the listed gaps: %.1f us" % tot)
