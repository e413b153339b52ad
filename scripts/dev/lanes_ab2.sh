# development aid (GPU box): the metric step by enqueue size and lanes
for c in ${COMBOS:-"4096 4" "2048 8" "2048 4" "1024 16" "4096 4" "2048 8"}; do set -- $c; python bench.py --only cfg2x --no-production-legs --steps 8 --warmup 2 --launch-batch $1 --lanes $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('launch batch $1 lanes $2', round(d['ms_per_step'],2), 'ms,', '%.3e' % d['value'])"; done
