python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -8
python -m pytest tests/test_gpu_step.py -x -q -k "16bit or highres" 2>&1 | tail -8
