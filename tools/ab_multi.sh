#!/bin/bash
# Same-box A/B of several environment settings: tools/ab_multi.sh REPS "A=1 B=2" "A=3" ...  (alternating bench.py runs)
REPS=$1; shift
for i in $(seq $REPS); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['all_kernels_ms_per_step'],3), flush=True)"
  done
done
