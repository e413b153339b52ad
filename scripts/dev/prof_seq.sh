#!/bin/bash
# development aid (GPU box): kernel trace of the sequential mapping loop; period between successive prepare kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/seqprof -o seq -- python3 scripts/dev/seq_chain_only.py 600 ${1:-1} ${@:2} > gpurun_out/seqprof.log 2>&1
python3 - <<'PY'
import csv,glob,statistics
f=glob.glob('gpurun_out/seqprof/**/seq_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %6s avg %8.0f ns min %s max %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
f=glob.glob('gpurun_out/seqprof/**/seq_kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
prep=[int(r["Start_Timestamp"]) for r in rows if "prepare_kernel" in r["Kernel_Name"]]
per=[b-a for a,b in zip(prep,prep[1:])]
per.sort()
print("prepare-to-prepare period ns: median %d mean %d p10 %d p90 %d max %d" % (statistics.median(per), statistics.mean(per), per[len(per)//10], per[len(per)*9//10], per[-1]))
# busy time per step
busy=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows if "ym::" in r["Kernel_Name"])/max(1,len(prep))
print("kernel busy per step ns: %d" % busy)
big=[(b-a) for a,b in zip(prep,prep[1:]) if b-a>80000]
print("periods > 80 us:", len(big), big[:10])
PY
find gpurun_out/seqprof -size +5M -delete
