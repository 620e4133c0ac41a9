python -m pytest tests/test_gpu_ops.py -x -q -k "wgrad_rows or conv_backward" 2>&1 | tail -3
echo "stagger"; ROWS_WGS=256 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu
echo "no stagger"; FALNET_WR_ABL=20 ROWS_WGS=256 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu | awk -F'|' '{print $1 "|" $3}'
python tools/wr_stamps.py 64 64 256 512 up 2>&1 | grep -v amdgpu
python tools/wr_stamps.py 128 128 64 128 2>&1 | grep -v amdgpu
