#!/bin/bash
# per-kernel durations of the metric workload in its two forms (one query object per enqueue / one query per item), one lane
tag=${1:-r05b}
out=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for form in one_query distinct_queries; do
  rm -rf $out/${tag}_${form}_trace
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${form}_trace -o ${tag} -- python3 bench.py --only cfg2x --no-production-legs --only-headline --headline $form --lanes 1 --steps 3 --warmup 1 > $out/${tag}_${form}_line.json 2> $out/${tag}_${form}_trace.log
  f=$(find $out/${tag}_${form}_trace -name "*kernel_stats.csv" | head -1)
  echo "== $form"; head -14 "$f" | cut -c1-200
  find $out/${tag}_${form}_trace -name "*kernel_trace.csv" -delete
done
