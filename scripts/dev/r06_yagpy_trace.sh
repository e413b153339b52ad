#!/bin/bash
# round 6: kernel trace of the cfg2x_yagpy leg (which kernels an enqueue of the reference's Python semantics runs, and for how long)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/yagt -o yagt -- python3 bench.py --only cfg2x --only-headline --legs cfg2x_yagpy --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r06_yagt_line.json 2> gpurun_out/r06_yagt.err
echo "rc $?"
python3 scripts/dev/yag_trace_summary.py | tee gpurun_out/r06_yagpy_kernels.md
find gpurun_out/yagt -name "*.csv" -size +20M -delete
