set -x
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/pytest_r2d.txt; cat gpurun_out/pytest_r2d.txt
