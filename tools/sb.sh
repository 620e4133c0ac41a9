FALNET_WGRAD_ROWS_WGS=128 FALNET_AUTOTUNE=0 python tools/slab_bytes.py 2>&1 | grep -v amdgpu | tail -45
