#!/bin/bash
cd "$(dirname "$0")/.."
echo "--- as built (launch_bounds 4 waves/SIMD for the 16-row tile kernel) ---"
python -m pytest tests/test_warp_gpu.py -m gpu -x -q 2>&1 | tail -2
python tools/quick_warp.py smooth 2>&1 | tail -3
CERB_EXTRA_HIPCC_FLAGS=-DCERB_TILE16_WPS=3 python -m cerberusnet_amd.build --force > /dev/null 2>&1
echo "--- launch_bounds 3 ---"
python tools/quick_warp.py smooth 2>&1 | tail -3
CERB_EXTRA_HIPCC_FLAGS=-DCERB_TILE16_WPS=2 python -m cerberusnet_amd.build --force > /dev/null 2>&1
echo "--- launch_bounds 2 ---"
python tools/quick_warp.py smooth 2>&1 | tail -3
