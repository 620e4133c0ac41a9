#!/bin/bash
# Same-box A/B of environment settings with a warmed scratch autotune cache, 10-step windows and the stream self-test in every line:
#   tools/ab_env2.sh REPS OUT "A=1" "A=0" ...
REPS=$1; OUT=$2; shift; shift
export FALNET_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/ab_cache.json
for cfg in "$@"; do env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1; done
for i in $(seq $REPS); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); w=d.get('ms_per_step_10step_windows',{}); t=d['config'].get('stream_selftest',{})
print('$cfg', round(d['value'],1), round(d['ms_per_step'],3), 'windows', w.get('min'), w.get('median'), w.get('max'), 'replaced', t.get('streams_replaced'), 'worst_pair', t.get('worst_pair_ms'), flush=True)" >> $OUT
  done
done
cat $OUT
