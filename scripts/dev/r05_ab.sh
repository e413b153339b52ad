#!/bin/bash
# A/B of debug-option sets on the metric workload: for every argument ("name=opt:val,opt:val" or "name=") one bench run of the
# cfg2x leg; prints ms per step, the region kernel's live duration and the per-enqueue GPU time.  Extra bench flags: $BENCH_FLAGS
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  name=${spec%%=*}; opts=${spec#*=}
  YM_BENCH_OPTS="$opts" timeout 600 python bench.py --only cfg2x --no-production-legs $BENCH_FLAGS > $out/ab_$name.json 2> $out/ab_$name.err || { echo "$name FAILED"; tail -5 $out/ab_$name.err; }
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % name).read().strip().splitlines()[-1])
    r = d["roofline"]
    dq = d["config"]["by_config"].get("cfg2x_distinct_queries") or d["config"]["by_config"].get("cfg2x_one_query") or {}
    print("%-14s step %.2f ms (min %.2f med %.2f max %.2f)  kernel %.0f us  call %.0f us  other-form step %s" % (
        name, d["ms_per_step"], d["ms_per_step_spread"]["min"], d["ms_per_step_spread"]["median"], d["ms_per_step_spread"]["max"],
        r["kernel_us"], r["call_us_gpu"], ("%.2f" % dq["ms_per_step"]) if dq else "-"))
except Exception as e:
    print(name, "no line:", e)
PY
done
