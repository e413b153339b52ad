#!/bin/bash
# round 6: the fine pass of the reference's Python semantics by rows (yag_fine_kernel): parity, soak, timing of the cfg2x_yagpy leg
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "yag or golden or sums" 2>&1 | tail -5
timeout 600 python3 scripts/dev/r06_yag_soak.py 0 10 60 2>&1 | tail -3
bash scripts/dev/r06_yagpy_trace.sh 2>&1 | tail -30
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_yagt_line.json").read().strip().splitlines()[-1])
y = d["config"]["by_config"]["cfg2x_yagpy"]
print({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("us_per_enqueue", "hypotheses_per_s", "items_that_fell_back")}) for k, v in y.items() if k != "what"})
PY
