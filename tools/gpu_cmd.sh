timeout 900 python tools/peak_sweep.py --kernels half2 --lengths 512 2>&1 | grep -v amdgpu.ids | cut -c1-420
timeout 900 python tools/peak_sweep.py --kernels half2 --lengths 512 2>&1 | grep -v amdgpu.ids | cut -c1-420
