#!/bin/bash
# round 6: does a high-priority second stream (the pair lists' launch) change the step?  A/B in one run, twice each.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for pr in 1 0; do
    YM_SIDE_STREAM_PRIORITY=$pr python3 bench.py --only cfg2x --no-production-legs --only-headline --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['ms_per_step_spread']
print('priority $pr: %.3f ms per step (min %.2f median %.2f max %.2f)' % (d['ms_per_step'], s['min'], s['median'], s['max']))"
  done
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pr in 1 0; do
  rm -rf gpurun_out/ovl
  YM_SIDE_STREAM_PRIORITY=$pr rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl -o ovl -- python3 bench.py --only cfg2x --no-production-legs --only-headline --steps 8 --warmup 2 > /dev/null 2> /dev/null
  echo "priority $pr:"; python3 scripts/dev/step_overlap.py | grep -E "bin_kernel|correlate_region|4 or more"
done
rm -rf gpurun_out/ovl
