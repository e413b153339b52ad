#!/bin/bash
# Run on the GPU box: the counter passes of a round for the three workloads whose kernels dominate a BASELINE config -- the
# metric's batch (cfg2x: correlate_region_kernel, raster, finish, cells), the loop-closure batch (cfg4: gather_kernel) and the
# stress match (cfg5: correlate_kernel<2,16,1>, select_relax_kernel) --, each counter group in its own rocprofv3 run, nothing else
# traced, copy kernels excluded; plus one kernel-trace pass per workload for the durations.  Summarised on the box
# (scripts/summarise_counters.py: the CSVs are too large to travel) into gpurun_out/<tag>_counters.json.
#   scripts/profile_pmc.sh r05p
tag=${1:-r05p}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in cfg2x cfg4 cfg5; do
  extra=""; [ $wl = cfg2x ] && extra="--no-production-legs --only-headline"; [ $wl = cfg4 ] && extra="--no-production-legs"
  only=${2:-}; [ -n "$only" ] && [ "$only" != $wl ] && continue   # (second argument: one workload only)
  rm -rf $out/${tag}_${wl}_trace
  timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${wl}_trace -o ${tag} -- python3 bench.py --only $wl $extra --lanes 1 --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_${wl}_trace.log
  for spec in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "l2:TCC_HIT_sum TCC_MISS_sum" "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
              "sq:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
              "sq2:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    name=${spec%%:*}; ctr=${spec#*:}
    rm -rf $out/${tag}_${wl}_$name
    timeout 420 rocprofv3 --pmc $ctr --kernel-exclude-regex "rocclr|at::native" --output-format csv -d $out/${tag}_${wl}_$name -o ${tag} -- python3 bench.py --only $wl $extra --lanes 1 --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_${wl}_$name.log
  done
done
python3 scripts/summarise_counters.py ${tag} > $out/${tag}_counters.log 2>&1
cat $out/${tag}_counters.log
find $out -name "*counter_collection.csv" -size +2M -delete
find $out -name "*kernel_trace.csv" -delete
