#!/bin/bash
# same-box A/B of the data-parallel step's collective placement (run on the GPU box): single GPU vs world-1 RCCL group with the bucket all-reduces
# on the backend's own stream (FALNET_COMM_ON_AUX=0, rounds 4-5) vs as sync collectives on the auxiliary stream (default, round 6)
out=gpurun_out/r06_ab_dist.txt
: > $out
line() { python3 - "$1" "$2" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{"metric')]
d=json.loads(l[-1]); a=d.get('allreduce',{})
print(sys.argv[2], round(d['value'],1), round(d['ms_per_step'],3), 'exposed_ms', a.get('exposed_ms'), 'without_collective', a.get('ms_per_step_without_collective'),
      'all3x200', (a.get('stream_selftest') or d['config'].get('stream_selftest') or {}).get('all_streams_3x200us_ms'), 'coll_beside_spin', (a.get('stream_selftest') or {}).get('collective_beside_spin_ms'))
PY
}
for r in 1 2 3; do
  python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-trajectory > gpurun_out/_s.log 2>&1; line gpurun_out/_s.log "single" >> $out
  FALNET_FORCE_DIST=1 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-trajectory > gpurun_out/_d1.log 2>&1; line gpurun_out/_d1.log "world-1 RCCL, collectives on aux (default)" >> $out
  FALNET_AB=1 FALNET_COMM_ON_AUX=0 FALNET_FORCE_DIST=1 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-trajectory > gpurun_out/_d0.log 2>&1; line gpurun_out/_d0.log "world-1 RCCL, collectives on the backend's stream (r5)" >> $out
done
cat $out
