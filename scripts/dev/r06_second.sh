#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06b
python -m pytest tests -m gpu -x -q -k "bulk_scan or scan_pool or threads" 2>&1 | tail -5
python scripts/dev/r06_create_time.py 2>/dev/null | tee gpurun_out/r06b/create_time.json
python bench.py --steps 20 --warmup 5 --only cfg2x > gpurun_out/r06b/bench.json 2> gpurun_out/r06b/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06b/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), d.get("setup"))
for k in ("cfg2x_fresh_scans",):
    print(k, json.dumps(d["config"]["by_config"].get(k), indent=None)[:1500])
print("errors", d.get("leg_errors"))
PY
