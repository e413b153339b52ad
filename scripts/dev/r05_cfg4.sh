#!/bin/bash
# cfg4 (loop-closure batch, gather correlate) with several library builds:  scripts/dev/r05_cfg4.sh base ga4 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  lib=$PWD/yag_slam_amd/libyagmatch_$n.so; [ "$n" = base ] && lib=$PWD/yag_slam_amd/libyagmatch.so
  YM_LIB_PATH=$lib timeout 600 python bench.py --only cfg4 > gpurun_out/cfg4_$n.json 2> gpurun_out/cfg4_$n.err || tail -3 gpurun_out/cfg4_$n.err
  python3 - $n <<'PY'
import json, sys
d = json.loads(open("gpurun_out/cfg4_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]["by_config"]["cfg4_loop_closure_batch"]
print("%-8s ms_per_query %.3f  first-use %.3f  one-shot %.3f  gather kernel %.0f us" % (sys.argv[1], c["ms_per_query"], c["ms_per_query_first_use_of_the_slots"], c["one_shot_ms_incl_results"], (c.get("roofline") or {}).get("kernel_us", 0)))
PY
done
