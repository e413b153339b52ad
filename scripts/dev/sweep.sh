#!/bin/bash
# development aid: bench.py --only cfg2x over a list of "name=args" variants (prints step ms and the profiled kernel's us)
for v in "$@"; do
  name="${v%%=*}"; args="${v#*=}"
  python bench.py --only cfg2x --steps 5 --warmup 2 $args 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$name', round(d['ms_per_step'],3), round(d['roofline']['kernel_us'],1))"
done
