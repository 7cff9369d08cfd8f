timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8
for v in 0 100000; do echo "LANES8_MAX_SUBJECT=$v"; CUDASW4_AMD_LANES8_MAX_SUBJECT=$v timeout 900 python tools/peak_sweep.py --kernels half2,dpxs16 --lengths 128,256,512 2>&1 | grep -v amdgpu.ids | cut -c1-400; done
