#!/bin/bash
# Run on the GPU box (through gpurun): every profile pass of a round, each in its own rocprofv3 run.
#   scripts/profile_round.sh r02p
# kernel trace of the default bench command; of one cfg2 match; of the stress match; PMC passes (FETCH_SIZE | L2 hits and
# misses | vector-L1 accesses), counters never combined with other tracing.
tag=${1:-r02p}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -o ${tag} -- python3 bench.py --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_single -o ${tag} -- python3 scripts/quick_time.py > $out/${tag}_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stress -o ${tag} -- python3 scripts/stress_time.py > $out/${tag}_stress.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -o ${tag} -- python3 bench.py --only cfg2x --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_fetch.log
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${tag}_l2 -o ${tag} -- python3 bench.py --only cfg2x --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_l2.log
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $out/${tag}_tcp -o ${tag} -- python3 bench.py --only cfg2x --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_tcp.log
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
find $out -name "*${tag}*" -size +20M -delete
ls $out | grep ${tag}
