python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/quick_warp.py smooth 2>&1 | tail -3
python tools/quick_warp16.py 2>&1 | tail -8
python bench.py 2>&1 | tail -1 > gpurun_out/bench_now.json
python bench.py --dtype f16 --width 2048 --height 1024 2>&1 | tail -1 > gpurun_out/bench_c5.json
python - <<'PY'
import json
for f in ('gpurun_out/bench_now.json','gpurun_out/bench_c5.json'):
    s=open(f).read(); d=json.loads(s[s.index('{'):])
    print(d['value'], d['ms_per_step'])
    print({k:v['us'] for k,v in sorted(d['roofline']['per_kernel'].items())})
PY
