#!/bin/bash
# A/B of library builds on the metric workload (one query per enqueue, one lane -> the region kernel's own duration, then the 4-lane step):
#   scripts/dev/r05_libs.sh name1 name2 ...   (yag_slam_amd/libyagmatch_<name>.so; "base" = the product library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in "$@"; do
  lib=$PWD/yag_slam_amd/libyagmatch_$n.so; [ "$n" = base ] && lib=$PWD/yag_slam_amd/libyagmatch.so
  BENCH_FLAGS="--headline one_query --no-distinct-queries --lanes 1 --steps 6 --warmup 2" YM_LIB_PATH=$lib scripts/dev/r05_ab.sh ${n}_1lane= 2>&1 | tail -1
  BENCH_FLAGS="--headline one_query --no-distinct-queries" YM_LIB_PATH=$lib scripts/dev/r05_ab.sh ${n}_4lane= 2>&1 | tail -1
done
