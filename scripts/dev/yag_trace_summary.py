"""Run on the GPU box after `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/yagt -o yagt -- python3 bench.py --only cfg2x
--only-headline --legs cfg2x_yagpy --steps 1 --warmup 0 --no-cpu-baseline`: the kernels of ONE enqueue of the cfg2x_yagpy leg (4096 matches in the
reference's Python semantics) in its two forms -- coarse sums from the production correlate kernel (items yag_lattice_kernel proved regular), and the
pair-by-pair kernel -- as a markdown table (kept as profiles/r06_yagpy_kernels.md).  An enqueue = the launches from one cells_kernel to the next."""
import csv, glob, collections
f = glob.glob("gpurun_out/yagt/**/yagt_kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "ym::" not in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "cells_kernel" in r[2]]  # the first kernel of an enqueue of a batch (points cached)
enq = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    seg = rows[a:b]
    if not any("yag_setup_kernel" in r[2] for r in seg):
        continue  # (an enqueue of the Karto-semantics legs)
    # what follows a drain (another leg) is cut at the first gap of 2 ms
    cut = len(seg)
    for j in range(1, len(seg)):
        if seg[j][0] > max(s[1] for s in seg[:j]) + 2000000:
            cut = j
            break
    enq.append(seg[:cut])
forms = {"production": [e for e in enq if any("yag_lattice_kernel" in r[2] for r in e)], "pairwise": [e for e in enq if not any("yag_lattice_kernel" in r[2] for r in e)]}
for label, es in forms.items():
    es = [e for e in es if len(e) > 3]
    if not es:
        continue
    es = es[len(es) // 2:]  # the later enqueues: warm
    per, cnt = collections.Counter(), collections.Counter()
    wall = 0
    for e in es:
        wall += max(r[1] for r in e) - e[0][0]
        for s, t, n in e:
            per[n] += t - s; cnt[n] += 1
    print("\n**%s** (%d enqueues, %.2f ms each from the first kernel's start to the last one's end)\n" % (label, len(es), wall / len(es) * 1e-6))
    print("| kernel | launches per enqueue | us per enqueue |\n|---|---|---|")
    for n, v in per.most_common(14):
        print("| `%s` | %.1f | %.0f |" % (n, cnt[n] / len(es), v / len(es) * 1e-3))
