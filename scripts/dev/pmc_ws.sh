#!/bin/bash
# development aid (run on the GPU box): counter passes over the region correlate in both forms
#   scripts/dev/pmc_ws.sh tag "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" ...
tag=$1; shift
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctr in "$@"; do
  for form in 0 1; do
    rm -rf $out/${tag}_f${form}_pmc$i
    YM_DEVELOPMENT=1 YM_DEBUG_OPTIONS="32=$form" rocprofv3 --pmc $ctr --kernel-include-regex "correlate_region" --output-format csv -d $out/${tag}_f${form}_pmc$i -o ${tag} -- python3 bench.py --only cfg2x --no-production-legs --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_f${form}_pmc$i.log
    echo "form $form:"; python3 scripts/pmc_kernel.py $out/${tag}_f${form}_pmc$i region
  done
  i=$((i+1))
done
