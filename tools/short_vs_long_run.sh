#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe first : steps 20 ->', d['value'], d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --probe-after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe after : steps 20 ->', d['value'], d['ms_per_step'])"
done
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe first : steps 200 ->', d['value'], d['ms_per_step'])"
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline --probe-after 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe after : steps 200 ->', d['value'], d['ms_per_step'])"
