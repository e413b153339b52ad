"""Development aid: mean of each counter per kernel from rocprofv3 counter_collection CSVs under a directory."""
import collections, csv, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if len(sys.argv) < 3 or sys.argv[2] in k:
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    for c in sorted(d[k]):
        v = sorted(d[k][c])
        v = v[len(v) // 4:] if len(v) > 3 else v
        print("%-40s %-28s %.4g (n=%d)" % (k[:40], c, sum(v) / len(v), len(v)))
