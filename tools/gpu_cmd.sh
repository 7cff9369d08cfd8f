set -x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_r2b.txt; cat gpurun_out/pytest_r2b.txt
timeout 900 python tools/peak_sweep.py --lengths 512 --kernels half2,dpxs32 --json gpurun_out/peak_r2b.json 2>&1 | tail -5
