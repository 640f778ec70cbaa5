#!/bin/bash
cd "$(dirname "$0")/.."
enc() { echo $(( $1 + $2 * 256 + $3 * 65536 )); }
for combo in "-1 0 0" "1 0 0" "2 0 0" "3 0 0" "4 0 0" "5 0 0" "7 0 0" "10 0 0" "12 0 0" "0 3 0" "3 0 3" "6 2 0" "6 0 2"; do
  set -- $combo
  if [ "$1" = "-1" ]; then v=-1; else v=$(enc $1 $2 $3); fi
  echo "--- q1=$1 q2=$2 q3=$3 ---"; CERB_OPT=warp_stagger=$v python tools/quick_warp.py smooth 2>&1 | tail -3 | sed -E 's/fwd .*bwd ctx/bwd ctx/' | cut -c1-60
done
