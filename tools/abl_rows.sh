python -m pytest tests/test_gpu_inference.py -x -q -k "generated_pngs or native_size or entry_scripts" 2>&1 | tail -8
python -m pytest tests/test_gpu_step.py -x -q -k "stage2" 2>&1 | tail -4
