#!/bin/bash
# Run on the GPU box: the counter passes of a round, each in its own rocprofv3 run, nothing else traced (copy kernels excluded
# to keep the CSVs small): FETCH_SIZE | L2 hits and misses | vector-L1 accesses | VALU / LDS issue and LDS busy cycles | where the
# waves' time goes (waiting at s_waitcnt / barriers, issue stalls, issuing) and scalar instructions
tag=${1:-r03p}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "fetch:FETCH_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" "sq:SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "sq2:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES"; do
  name=${spec%%:*}; ctr=${spec#*:}
  rm -rf $out/${tag}_$name
  rocprofv3 --pmc $ctr --kernel-exclude-regex "rocclr|at::native" --output-format csv -d $out/${tag}_$name -o ${tag} -- python3 bench.py --only cfg2x --no-production-legs --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_$name.log
  ls -la $out/${tag}_$name
done
# the counter CSVs are too large to travel: summarise here, keep the summaries
python3 scripts/summarise_profiles.py ${tag} ${2:-4096} $out/${tag}_summary > $out/${tag}_summary.log 2>&1
find $out -name "*counter_collection.csv" -size +5M -delete
