#!/bin/bash
# Regenerate the persistent autotune cache on the GPU box over every bench workload / dtype, then take the round's profile with it.
# usage (on the box): bash tools/regen_cache.sh r02
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export FALNET_AUTOTUNE_CACHE=$out/autotune_cache.json
rm -f $FALNET_AUTOTUNE_CACHE
for cfg in "stage1 bf16" "stage1 f16" "stage1 f32" "stage2 bf16" "stage2 f16" "highres f16" "highres bf16"; do
  set -- $cfg
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 3 --no-cpu-baseline --no-live-traffic 2>&1 | tail -1 > $out/bench_$1_$2.json
  python -c "import json,sys; d=json.load(open('$out/bench_$1_$2.json')); print('$cfg', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms')"
done
cp $FALNET_AUTOTUNE_CACHE fal_net_amd/autotune_cache.json
unset FALNET_AUTOTUNE_CACHE
python bench.py 2>&1 | tail -1 > $out/bench_default.json
cat $out/bench_default.json
