#!/bin/bash
# How often does a default bench run land on the slow mode?  tools/cliff_probe.sh REPS OUT "CFG" ...  (each run: ms/step, 10-step windows, stream self-test)
REPS=$1; OUT=$2; shift; shift
export FALNET_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/ab_cache.json
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1
for i in $(seq $REPS); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('$cfg', round(d['ms_per_step'],3), d.get('ms_per_step_10step_windows'), json.dumps(c.get('stream_selftest'))[:300], flush=True)" >> $OUT
  done
done
cat $OUT
