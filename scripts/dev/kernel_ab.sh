#!/bin/bash
# development aid (GPU box): kernel stats of the metric workload under two library builds (scripts/dev/ab/libA.so, libB.so
# through YM_LIB_PATH), alternating.   scripts/dev/kernel_ab.sh <kernel name substring>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in A B; do
  export YM_LIB_PATH=$GRAFT_REPO_ROOT/scripts/dev/ab/lib$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab$v -o ab -- python3 bench.py --only cfg2x --no-production-legs --no-cpu-baseline --steps 4 --warmup 2 > /dev/null 2>&1
  echo "== $v"
  grep "$1" gpurun_out/ab$v/ab_kernel_stats.csv | sed "s/\"[^\"]*\"/K/" | cut -d, -f1-4
  find gpurun_out/ab$v -size +1M -delete
done
done
