#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_warp_gpu.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for o in -1 0; do
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline --no-traffic --option warp_stagger=$o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']
print('warp_stagger=$o: %.1f pairs/s %.4f ms | warp_bwd in-step L1 %.2f L2 %.2f L3 %.2f | hot L1 %.2f L2 %.2f L3 %.2f' % (d['value'], d['ms_per_step'], r['warp_bwd_L1']['us_in_step'], r['warp_bwd_L2']['us_in_step'], r['warp_bwd_L3']['us_in_step'], r['warp_bwd_L1']['us_hot'], r['warp_bwd_L2']['us_hot'], r['warp_bwd_L3']['us_hot']))"
done; done
