timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "process_group or two_ranks" 2>&1 | tail -15
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
