#!/bin/bash
# round 6: the default bench line with the round's counters in the tree (roofline not stale), the one-rank RCCL dry run of --gpus, a short soak
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
python3 bench.py > gpurun_out/r06z_bench.json 2> gpurun_out/r06z_bench.err; echo "bench rc $?"
YM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 bench.py --gpus 1 --steps 5 --no-cpu-baseline > gpurun_out/r06z_dist_dryrun.json 2> gpurun_out/r06z_dist_dryrun.err; echo "dry run rc $?"
bash scripts/dev/r06_soak.sh 619 | tail -3
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06z_bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.4g ms/step %.3f" % (d["value"], d["ms_per_step"]), "roofline", {k: r.get(k) for k in ("bound", "frac", "stale", "hbm_frac", "kernel_us")})
print("errors", d.get("leg_errors"))
PY
