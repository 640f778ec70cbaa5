python -m pytest tests/test_corr_gpu.py -x -q -k "mfma or half or dtypes or config5" 2>&1 | tail -3
python tools/quick_corr16.py f16 2>&1 | tail -9
