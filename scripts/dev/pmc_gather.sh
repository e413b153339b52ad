#!/bin/bash
# development aid (run on the GPU box): rocprofv3 counter passes over scripts/dev/gather_sweep.py (one setting), gather kernels only
#   scripts/dev/pmc_gather.sh tag "B [loop] na,parts,lds" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
tag=$1; shift
what=$1; shift
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctr in "$@"; do
  rm -rf $out/${tag}_pmc$i
  rocprofv3 --pmc $ctr --kernel-include-regex "gather_kernel" --output-format csv -d $out/${tag}_pmc$i -o ${tag} -- python3 scripts/dev/gather_sweep.py $what > /dev/null 2> $out/${tag}_pmc$i.log
  python3 scripts/pmc_kernel.py $out/${tag}_pmc$i gather_kernel
  i=$((i+1))
done
