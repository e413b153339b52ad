#!/bin/bash
# round 5, the state of HEAD: smoke, the whole GPU suite, the experimental build's forms, then every profile of the round
tag=${1:-r05z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
YM_LIB_PATH=$PWD/yag_slam_amd/libyagmatch_exp.so python -m pytest tests -m gpu -x -q -k "region_correlate or cfg2_batch_512 or loop" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -2
scripts/profile_round.sh $tag 2>&1 | tail -3
scripts/profile_pmc.sh $tag 2>&1 | tail -5
