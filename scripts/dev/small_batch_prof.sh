#!/bin/bash
# development aid (GPU box): kernel stats of a small batch enqueue
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sb -o sb -- python3 scripts/dev/small_batch.py "$@" > gpurun_out/sb.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/sb/**/sb_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ym::" in r["Name"]: print("%-75s calls %5s avg %8.1f us" % (r["Name"][:75], r["Calls"], float(r["AverageNs"])/1e3))
PY
find gpurun_out/sb -size +2M -delete
