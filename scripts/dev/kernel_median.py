"""Development aid: median duration per kernel from a rocprofv3 kernel trace CSV (us); large launches only."""
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    big = v[len(v) // 2:]  # (the self-check's single matches and the like are the small half, if there are any)
    if "ym::" in k:
        print("%-46s n=%4d median of the upper half %9.1f us" % (k[:46], len(v), statistics.median(big)))
