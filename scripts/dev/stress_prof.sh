#!/bin/bash
# development aid (GPU box): kernel stats of the stress match (BASELINE configs[4])
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stressprof -o st -- python3 scripts/stress_time.py > gpurun_out/stressprof.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/stressprof/**/st_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ym::" in r["Name"] and "structure" not in r["Name"]: print("%-75s calls %5s avg %8.1f us min %8.1f" % (r["Name"][:75], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
find gpurun_out/stressprof -size +1M -delete
