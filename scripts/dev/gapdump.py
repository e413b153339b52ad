"""Development aid: what the stream does around a call boundary (after argbest_kernel) in a rocprofv3 trace: kernels and, when
rs_memory_copy_trace.csv lies next to the kernel trace, memory copies; times in us relative to the end of argbest."""
import csv, os, sys
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in csv.DictReader(open(sys.argv[1]))]
mc = os.path.join(os.path.dirname(sys.argv[1]), os.path.basename(sys.argv[1]).replace("kernel_trace", "memory_copy_trace"))
if os.path.exists(mc):
    ev += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"][12:]) for r in csv.DictReader(open(mc))]
ev.sort()
idx = [i for i, r in enumerate(ev) if "argbest" in r[2]]
for i in idx[-3:-1]:
    t0 = ev[i][1]
    for s, e, n in ev[max(0, i - 2):i + 8]:
        print("%9.1f %9.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, n))
    print()
