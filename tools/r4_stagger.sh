#!/bin/bash
cd "$(dirname "$0")/.."
enc() { echo $(( $1 + $2 * 256 + $3 * 65536 )); }
for combo in "0 0 0" "6 0 0" "6 0 6" "6 3 9" "8 4 12" "0 4 4" "0 6 6" "0 0 6" "4 8 12" "6 6 12" "8 0 8" "3 6 9" "8 2 10"; do
  set -- $combo
  echo "--- delays (x1024 cycles) q1=$1 q2=$2 q3=$3 ---"; CERB_OPT=warp_stagger=$(enc $1 $2 $3) python tools/quick_warp.py smooth 2>&1 | tail -3 | cut -c1-40,95-170
done
