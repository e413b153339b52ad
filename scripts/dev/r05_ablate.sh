#!/bin/bash
# what each part of a round costs correlate_region_kernel<8, true>: timing-only builds (-DYM_RG_ABLATE=bits, wrong sums) of the
# library, one bench run of the metric workload each (one query per enqueue, one lane: the kernel's own duration)
# build first, in the container:  for b in 1 2 4 6 7 3; do make -s -C yag_slam_amd/csrc OUT=../libyagmatch_ab$b.so FLAGS_EXTRA=-DYM_RG_ABLATE=$b; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BENCH_FLAGS="--headline one_query --no-distinct-queries --lanes 1 --steps 6 --warmup 2"
scripts/dev/r05_ab.sh full= 2>&1 | grep -v "^$"
for b in "$@"; do
  YM_LIB_PATH=$PWD/yag_slam_amd/libyagmatch_ab$b.so YM_BENCH_SKIP_CHECK=1 scripts/dev/r05_ab.sh ablate$b= 2>&1 | tail -1
done
