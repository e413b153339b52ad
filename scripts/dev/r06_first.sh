#!/bin/bash
# round 6, first measurement: the new GPU tests, then the bench with the new legs (fresh scans, yagpy)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q -k "bulk_scan or pair_lists or yagpy" 2>&1 | tail -15 > gpurun_out/r06a/tests.log
cat gpurun_out/r06a/tests.log
python bench.py --steps 20 --warmup 5 --only cfg2x > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
echo "bench rc $?"
tail -5 gpurun_out/r06a/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06a/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("setup"))
for k in ("cfg2x_one_query", "cfg2x_fresh_scans", "cfg2x_yagpy", "cfg2x_fresh_query"):
    print(k, json.dumps(d["config"]["by_config"].get(k), indent=None)[:1500])
print("errors", d.get("leg_errors"))
PY
