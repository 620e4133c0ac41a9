for cfg in "FALNET_DISABLE_PATCH=1" "FALNET_PATCH_KCB=128" "FALNET_PATCH_KCB=64" "FALNET_PATCH_KCB=0"; do
  echo "== $cfg"; env $cfg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2)); print({k:v for k,v in list(d['kernel_breakdown_ms_per_step'].items())[:5]})"
done
