#!/bin/bash
# round 6: tests/soak.py for many seeds (random call sequences on ONE long-lived matcher against a matcher that forgets everything),
# after the round's host changes: pair lists per angle block, a query's projection cached from its second use, event-based block
# recycling, bulk scan creation, the tall-tile pattern
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06soak
fail=0; n=0
for seed in $(seq 600 1 ${1:-659}); do
  python3 scripts/dev/soak_calls.py $seed 300 > gpurun_out/r06soak/s$seed.log 2>&1 || { fail=$((fail+1)); echo "seed $seed FAILED"; tail -5 gpurun_out/r06soak/s$seed.log; }
  n=$((n+1))
done
echo "soak: $n seeds x 300 calls, $fail failed"
