#!/bin/bash
# round 6, the state of HEAD: smoke, the whole GPU suite (product + experimental build), every profile of the round, the step overlap
tag=${1:-r06z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed"
YM_LIB_PATH=$PWD/yag_slam_amd/libyagmatch_exp.so python -m pytest tests -m gpu -x -q -k "region_correlate or cfg2_batch_512 or loop" 2>&1 | grep -E "passed|failed"
scripts/profile_round.sh $tag 2>&1 | tail -3
scripts/profile_pmc.sh $tag 2>&1 | tail -5
rm -rf gpurun_out/ovl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl -o ovl -- python3 bench.py --only cfg2x --no-production-legs --only-headline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/${tag}_ovl.log
python3 scripts/dev/step_overlap.py > gpurun_out/${tag}_step_overlap.md 2>&1
rm -rf gpurun_out/ovl
python3 scripts/dev/seq_stamps.py 400 > gpurun_out/${tag}_seq_stamps.txt 2>&1
python3 scripts/dev/r06_create_time.py > gpurun_out/${tag}_create_time.json 2>/dev/null
tail -3 gpurun_out/${tag}_step_overlap.md
