#!/bin/bash
cd "$(dirname "$0")/.."
for o in 0 1 2 3; do echo "--- warp_no_lists=$o (bit0: scan instead of lists, bit1: range-major tiles) ---"; CERB_OPT=warp_no_lists=$o python tools/quick_warp.py smooth 2>&1 | tail -3; done
