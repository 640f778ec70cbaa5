python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/quick_corr16.py f16 bf16 2>&1 | tail -16
python bench.py --dtype f16 --width 2048 --height 1024 2>&1 | tail -1 > gpurun_out/bench_c5.json
python bench.py --dtype bf16 2>&1 | tail -1 > gpurun_out/bench_bf16.json
python - <<'PY'
import json
for f in ('gpurun_out/bench_c5.json','gpurun_out/bench_bf16.json'):
    s=open(f).read(); d=json.loads(s[s.index('{'):])
    print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
    print({k:v['us'] for k,v in sorted(d['roofline']['per_kernel'].items())})
PY
