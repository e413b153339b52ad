#!/bin/bash
# A/B of library builds on BOTH forms of the metric workload (four lanes): scripts/dev/r05_libs2.sh base h77 ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  lib=$PWD/yag_slam_amd/libyagmatch_$n.so; [ "$n" = base ] && lib=$PWD/yag_slam_amd/libyagmatch.so
  BENCH_FLAGS="" YM_LIB_PATH=$lib scripts/dev/r05_ab.sh ${n}= 2>&1 | tail -1
done
