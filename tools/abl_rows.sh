mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_step.py -x -q -s -k "trains_like" 2>&1 | grep -v Warning | grep "f32\|assert\|Error" | head -12
python tools/trajectory.py --steps 200 --height 128 --width 256 --batch 4 --pool 8 2>&1 | tail -1 > gpurun_out/r02/trajectory_200.json; cat gpurun_out/r02/trajectory_200.json
