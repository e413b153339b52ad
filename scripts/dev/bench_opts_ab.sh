for o in ${AB_LIST:-"" "30:64" "" "30:64"}; do YM_BENCH_OPTS=$o python bench.py --only cfg2x --no-production-legs --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['ms_per_step'], d['roofline']['kernel_us'])"; done
