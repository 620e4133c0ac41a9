#!/bin/bash
for i in 1 2 3; do for kv in "X=1" "FALNET_LABEL_VGG_MID=0"; do
  env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['config']['final_loss'])"
done; done
for s in 1 3 6 12; do python bench.py --steps $s --warmup 0 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $s', d['config']['final_loss'])"; FALNET_LABEL_VGG_MID=0 python bench.py --steps $s --warmup 0 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $s MID=0', d['config']['final_loss'])"; done
