python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/quick_warp.py smooth 2>&1 | tail -3
python bench.py 2>&1 | tail -1 > gpurun_out/bench_now.json
python - <<'PY'
import json
s=open('gpurun_out/bench_now.json').read(); d=json.loads(s[s.index('{'):])
print(d['value'], d['ms_per_step'])
for k,v in sorted(d['roofline']['per_kernel'].items()): print(k, v['us'])
PY
