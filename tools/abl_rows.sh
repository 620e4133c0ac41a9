python -m pytest tests/test_gpu_step.py tests/test_gpu_model.py tests/test_gpu_inference.py -x -q 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f16 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype f32 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload stage2 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload highres 2>&1 | tail -1 | cut -c1-200
cp fal_net_amd/autotune_cache.json gpurun_out/autotune_cache.json; wc -c gpurun_out/autotune_cache.json
