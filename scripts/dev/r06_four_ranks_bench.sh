#!/bin/bash
# round 6: bench.py --gpus 4 with four real ranks on one GPU (see r06_two_ranks_bench.sh), a quarter of the metric's batch per rank and one lane:
# uneven angle split (46 = 12 + 12 + 12 + 10), four shards of 1024 loop chains
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
YM_BENCH_WATCHDOG=500 YM_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 4 --steps 6 --warmup 2 --batch 4096 --lanes 1 --no-cpu-baseline --cfg3-scans 200 > gpurun_out/r06_four_ranks.json 2> gpurun_out/r06_four_ranks.err
echo "rc $?"
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_four_ranks.json") if l.startswith("{")][-1])
bc = d["config"]["by_config"]
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "errors", d.get("leg_errors"))
c4 = bc.get("cfg4_loop_closure_batch", {}); print("cfg4", {k: c4.get(k) for k in ("chains", "chains_per_gpu", "ms_per_query", "winner")})
c5 = bc.get("cfg5_stress", {}); print("cfg5", c5.get("split_by_angle"))
PY
grep -A12 "most recent call first" gpurun_out/r06_four_ranks.err | grep -v "^W1004" | head -20
