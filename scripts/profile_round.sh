#!/bin/bash
# Run on the GPU box (through gpurun): the kernel-trace passes of a round, each in its own rocprofv3 run, then the plain
# bench line.  The counter passes are scripts/profile_pmc.sh (counters are never combined with other tracing).
#   scripts/profile_round.sh r02p
# kernel trace of the default bench command; of one cfg2 match; of the stress match.
tag=${1:-r05p}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o ${tag} -- python3 bench.py --lanes 1 --no-cpu-baseline --no-production-legs --only-headline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_single -o ${tag} -- python3 scripts/quick_time.py > $out/${tag}_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stress -o ${tag} -- python3 scripts/stress_time.py > $out/${tag}_stress.log 2>&1
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# the exact code path of --gpus N (all-gather per step over the lanes, cfg4 sharding, cfg5 split by angle) with ONE rank over RCCL
YM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 bench.py --gpus 1 --steps 5 --no-cpu-baseline > $out/${tag}_dist_dryrun.json 2> $out/${tag}_dist_dryrun.err
find $out -path "*${tag}_*" -name "*kernel_trace.csv" -delete   # (only the per-kernel statistics travel)
find $out -name "*${tag}*" -size +20M -delete
ls $out | grep ${tag}
