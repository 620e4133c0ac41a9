#!/bin/bash
# Same-box A/B of environment settings with a warmed scratch autotune cache: tools/ab_cached.sh REPS OUT "A=1" "A=0" ...
# (one untimed filling run per setting, then alternating runs; the packaged cache is not touched)
REPS=$1; OUT=$2; shift; shift
export FALNET_AUTOTUNE_CACHE=$GRAFT_REPO_ROOT/gpurun_out/ab_cache.json
rm -f $FALNET_AUTOTUNE_CACHE
for cfg in "$@"; do env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2>&1; done
bash tools/ab_env.sh $REPS $OUT "$@"
