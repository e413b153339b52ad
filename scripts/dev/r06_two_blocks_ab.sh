#!/bin/bash
# round 6: does the STEP gain when the region correlate leaves a third of every CU to the other lanes' kernels?  (option 38: dynamic LDS the
# kernel is launched with and does not use -- 12 000 bytes: two blocks per CU instead of three, 38 KB of LDS and a third of the wave slots free)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for rep in 1 2; do
for opts in "" "38:12000" "38:6000"; do
  YM_BENCH_OPTS="$opts" timeout 300 python3 bench.py --only cfg2x --no-production-legs --only-headline --steps 40 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('opts [%s] ms/step %.3f spread %s kernel_us %.0f' % ('$opts', d['ms_per_step'], [round(d['ms_per_step_spread'][k],2) for k in ('min','median','max')], d['roofline']['kernel_us']))"
done; done
