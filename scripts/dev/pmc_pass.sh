#!/bin/bash
# development aid (run on the GPU box): one rocprofv3 counter pass per argument over `bench.py --only cfg2x`, copy kernels excluded
#   scripts/dev/pmc_pass.sh tag "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
tag=$1; shift
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctr in "$@"; do
  rm -rf $out/${tag}_pmc$i
  rocprofv3 --pmc $ctr --kernel-include-regex "region|raster" --output-format csv -d $out/${tag}_pmc$i -o ${tag} -- python3 bench.py --only cfg2x --steps 2 --warmup 1 > /dev/null 2> $out/${tag}_pmc$i.log
  python3 scripts/pmc_kernel.py $out/${tag}_pmc$i region
  i=$((i+1))
done
