python -m pytest tests/test_gpu_render.py tests/test_gpu_mix_fusion.py tests/test_gpu_captured_render.py -q -m gpu --tb=short 2>&1 | tail -12
python bench.py --no-train --no-cpu-baseline --no-sustained > gpurun_out/bench_r5d.json 2> gpurun_out/bench_r5d.err; tail -c 300 gpurun_out/bench_r5d.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r5d.json'))
print(d['ms_per_step'], d['roofline']['per_kernel_ms_per_step'])
for k,v in d['secondary'].items():
    print(k, v.get('ms_per_step'), v.get('error'), (v.get('roofline') or {}).get('per_kernel_ms_per_step'))
PY
