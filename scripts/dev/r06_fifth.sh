#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06e
( time python bench.py > gpurun_out/r06e/bench.json 2> gpurun_out/r06e/bench.err ) 2>&1 | tail -3
echo "bench rc $?"
tail -3 gpurun_out/r06e/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06e/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.3f steps %d" % (d["value"], d["ms_per_step"], d["steps"]), d.get("setup"), d["headline_workload"])
bc = d["config"]["by_config"]
for k in ("cfg2x_one_query", "cfg2x_trajectory", "cfg2x_fresh_scans"):
    v = bc.get(k) or {}
    print(k, {kk: vv for kk, vv in v.items() if kk not in ("what", "ms_per_step_spread")})
print("cfg4", {kk: vv for kk, vv in bc["cfg4_loop_closure_batch"].items() if kk in ("ms_per_query", "cfg4_shard_512")})
print("single", bc["cfg2_single_match"]["sync_us_per_match"], "cfg3", bc["cfg3_sequential_mapping"]["scan_matches_per_s"], "cfg5", bc["cfg5_stress"]["us_per_match"])
print("roofline", {k: d["roofline"].get(k) for k in ("stale", "frac", "kernel_us", "stale_because")})
print("vs_cpu", d.get("vs_cpu_baseline"), "errors", d.get("leg_errors"))
PY
