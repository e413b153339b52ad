#!/bin/bash
# round 6: bench.py's --gpus 2 code path with two REAL ranks on the one GPU of a test box (gloo, records staged through the host): a dry run of the
# N > 1 logic -- sharding of cfg4 by rank, the weak-scaling aggregation of the metric, barriers, cfg5 split by angle -- not a scaling measurement
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
YM_BENCH_WATCHDOG=${WATCHDOG:-400} YM_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --cfg3-scans 300 > gpurun_out/r06_two_ranks.json 2> gpurun_out/r06_two_ranks.err
echo "rc $?"; tail -3 gpurun_out/r06_two_ranks.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_two_ranks.json") if l.startswith("{")][-1])
bc = d["config"]["by_config"]
print("n_gpus", d["n_gpus"], "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "scaling", d["scaling"], "errors", d.get("leg_errors"))
print("collective:", d["config"].get("collective"))
c4 = bc.get("cfg4_loop_closure_batch", {}); print("cfg4", {k: c4.get(k) for k in ("chains", "chains_per_gpu", "ms_per_query", "winner", "collective")})
c5 = bc.get("cfg5_stress", {}); print("cfg5", c5.get("us_per_match"), c5.get("split_by_angle"))
PY
