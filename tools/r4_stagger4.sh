#!/bin/bash
cd "$(dirname "$0")/.."
for o in -1 0; do
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline --no-traffic --serial-directions --option warp_stagger=$o 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']['per_kernel']
print('serial, warp_stagger=$o: %.1f pairs/s %.4f ms' % (d['value'], d['ms_per_step']))
print('   hot:', ' '.join('%s %.1f' % (k, v['us_hot']) for k, v in sorted(r.items())))"
done
python tools/quick_warp.py smooth 2>&1 | tail -3 | cut -c1-175
