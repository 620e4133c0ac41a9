#!/bin/bash
# same-box whole-step A/B of alternative library builds: tools/ab_lib_bench.sh "<tag> <tag> ..." [reps]   ("base" = in-tree library)
tags=$1; REPS=${2:-2}
for i in $(seq $REPS); do
  for t in $tags; do
    if [ "$t" = base ]; then unset FALNET_LIB; else export FALNET_LIB=$PWD/fal_net_amd/libfalnet_hip_$t.so; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-live-traffic 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['all_kernels_ms_per_step'],3))"
  done
done
