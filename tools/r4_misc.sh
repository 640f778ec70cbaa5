#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_runtime.py -m gpu -q 2>&1 | tail -3
./tools/ubench/mfma_coissue
python tools/f2_bound_coarse.py 2>&1 | grep -v amdgpu.ids
