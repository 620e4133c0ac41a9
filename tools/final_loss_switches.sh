#!/bin/bash
# final loss of the 25-step bench run under each feature switch (numerical regression hunting)
for kv in "X=1" "FALNET_FUSED_BIAS=0" "FALNET_WGRAD_S2=0" "FALNET_LABEL_VGG_MID=0" "FALNET_COMPOSE_LOGITS=0" "FALNET_TAIL_BALANCE=0" "FALNET_PACK_AFTER_ADAM=0" "FALNET_HEAD_BWD_V1=1" "FALNET_WGRAD_CO2=0"; do
  env $kv python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['config']['final_loss'], round(d['value'],1))"
done
