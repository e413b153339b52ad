#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06d
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "--- experimental build"
YM_LIB_PATH=$PWD/yag_slam_amd/libyagmatch_exp.so python -m pytest tests -m gpu -x -q -k "region or cfg2_batch or pairs" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --only cfg2x --no-production-legs > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06d/bench.json").read().strip().splitlines()[-1])
print("value %.4g ms/step %.3f spread %s" % (d["value"], d["ms_per_step"], d["ms_per_step_spread"]))
k = d["config"]["by_config"].get("cfg2x_one_query")
print("one_query ms/step %.3f ratio %.4f" % (k["ms_per_step"], k["ratio_to_metric_line"]))
print("kernel_us", d["roofline"]["kernel_us"], "call_us_gpu", d["roofline"]["call_us_gpu"])
PY
