#!/bin/bash
# Same-box comparison of the round-5 tree (a git worktree at _r05/, its own library and committed autotune cache) and this tree: alternating default
# bench runs.  usage (on the GPU box): bash tools/ab_rounds.sh REPS OUT
REPS=${1:-4}; OUT=${2:-gpurun_out/r06_vs_r05_same_box.txt}
: > $OUT
line() { python3 -c "
import sys,json; d=json.loads([l for l in open('$1') if l.startswith('{\"metric')][-1]); w=d.get('ms_per_step_10step_windows',{})
print('$2', round(d['value'],1), round(d['ms_per_step'],3), 'windows', w.get('min'), w.get('median'), w.get('max'), flush=True)" >> $OUT; }
for i in $(seq $REPS); do
  (cd _r05 && python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > ../gpurun_out/_r05.log 2>&1); line gpurun_out/_r05.log "round-5 tree (995f362)"
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/_r06.log 2>&1; line gpurun_out/_r06.log "this tree"
done
cat $OUT
