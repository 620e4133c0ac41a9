#!/bin/bash
# Same-box A/B of environment settings on the default bench: tools/ab_env.sh REPS OUT "A=1" "A=0" ...  (alternating runs, one line each)
REPS=$1; OUT=$2; shift; shift
for i in $(seq $REPS); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', round(d['value'],1), round(d['ms_per_step'],3), d['config']['final_loss'], flush=True)" >> $OUT
  done
done
cat $OUT
