timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
timeout 900 python tools/short_query_sweep.py 2>&1 | grep -v amdgpu.ids
