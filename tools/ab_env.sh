#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh VAR A B [reps]  (alternating runs, pairs/s + ms/step + serial kernel ms)
VAR=$1; A=$2; B=$3; REPS=${4:-2}
for i in $(seq $REPS); do
  for v in $A $B; do
    export $VAR=$v
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['all_kernels_ms_per_step'],3))"
  done
done
