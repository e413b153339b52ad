#!/bin/bash
# Run on the GPU box: the three counter passes of scripts/profile_round.sh alone (copy kernels excluded to keep the CSVs small)
tag=${1:-r02p}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "fetch:FETCH_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum"; do
  name=${spec%%:*}; ctr=${spec#*:}
  rm -rf $out/${tag}_$name
  rocprofv3 --pmc $ctr --kernel-exclude-regex "rocclr|at::native" --output-format csv -d $out/${tag}_$name -o ${tag} -- python3 bench.py --only cfg2x --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_$name.log
  ls -la $out/${tag}_$name
done
