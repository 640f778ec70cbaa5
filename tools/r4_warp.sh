#!/bin/bash
# round 4: warp tests + timings of the list-based backward against the scan path
cd "$(dirname "$0")/.."
python -m pytest tests/test_warp_gpu.py -m gpu -x -q 2>&1 | tail -8
echo "--- lists (default) ---"; python tools/quick_warp.py smooth 2>&1 | tail -3
echo "--- scan (warp_no_lists) ---"; CERB_OPT=warp_no_lists python tools/quick_warp.py smooth 2>&1 | tail -3
./tools/ubench/lds_atomic_mask
