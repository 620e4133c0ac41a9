export FALNET_WGRAD_ROWS_WGS=128
python -m pytest tests/test_gpu_step.py -x -q 2>&1 | tail -2
for v in 256 64; do
FALNET_REDUCE_BLOCKS=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --launch-table gpurun_out/lt_$v.txt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RB=$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['all_kernels_ms_per_step'],3))"
grep "wgrad_reduce\|bias_grad_b" gpurun_out/lt_$v.txt
done
tools/ab_multi.sh 2 "FALNET_REDUCE_BLOCKS=256" "FALNET_REDUCE_BLOCKS=128" "FALNET_REDUCE_BLOCKS=64" "FALNET_REDUCE_BLOCKS=32"
