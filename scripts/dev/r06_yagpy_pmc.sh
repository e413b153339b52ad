#!/bin/bash
# round 6: counters of the kernels of the cfg2x_yagpy leg (three --pmc passes, nothing else traced), per launch of 4096 items
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for spec in "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "sq2:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${spec%%:*}; ctr=${spec#*:}
  rm -rf $out/yagpmc_$name
  timeout 420 rocprofv3 --pmc $ctr --kernel-include-regex "yag_|score_kernel" --output-format csv -d $out/yagpmc_$name -o yagpmc -- python3 bench.py --only cfg2x --only-headline --legs cfg2x_yagpy --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/yagpmc_$name.log
  echo "$name rc $?"
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/yagpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | launches seen | " + " | ".join(["VALU", "SALU", "LDS instr", "L1 line visits", "L1 requests", "wave cycles", "waiting", "FETCH KB", "WRITE KB"]) + " |")
for k, d in sorted(acc.items()):
    def big(name):  # the launches of 4096 items: the upper half of the values (the leg's first call is a single launch too)
        v = sorted(d.get(name, [0.0])); v = v[len(v) // 2:]; return sum(v) / len(v)
    n = len(d.get("SQ_INSTS_VALU", []))
    wc = big("SQ_WAVE_CYCLES")
    print("| `%s` | %d | %.3g | %.3g | %.3g | %.3g | %.3g | %.3g | %.2f | %.0f | %.0f |" % (k, n, big("SQ_INSTS_VALU"), big("SQ_INSTS_SALU"), big("SQ_INSTS_LDS"),
          big("TCP_TOTAL_CACHE_ACCESSES_sum"), big("TCP_TOTAL_ACCESSES_sum"), wc, big("SQ_WAIT_ANY") / wc if wc else 0, big("FETCH_SIZE"), big("WRITE_SIZE")))
PY
find $out -name "*counter_collection.csv" -size +2M -delete
