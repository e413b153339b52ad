#!/bin/bash
# development aid (GPU box): kernel trace of the sequential mapping loop
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/seqprof -o seq -- python3 scripts/dev/seq_stamps.py 600 > gpurun_out/seqprof.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/seqprof/**/seq_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-70s calls %6s avg %8.0f ns min %s max %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
find gpurun_out/seqprof -size +5M -delete
