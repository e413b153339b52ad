# development aid (GPU box): the metric step with 1 .. 4 lanes, alternating
for l in ${LANES:-2 3 1 4 2 3}; do python bench.py --only cfg2x --no-production-legs --steps 8 --warmup 2 --lanes $l 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $l', round(d['ms_per_step'],2), 'ms,', '%.3e' % d['value'])"; done
