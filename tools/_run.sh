python -m pytest tests/test_warp_gpu.py -x -q 2>&1 | tail -12
