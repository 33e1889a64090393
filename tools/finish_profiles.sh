#!/bin/bash
# after `gpurun -- bash tools/collect_profiles.sh` (which clears the raw directories first): distil the run into profiles/r4/
GRAFX_ROUND=r4 python tools/pmc_summary.py | tail -3
cp gpurun_out/profiles_raw/bench_cfg2.json profiles/r4/bench_cfg2.json
cp gpurun_out/profiles_raw/bench_cfg3.json profiles/r4/bench_cfg3.json
for c in cfg2 cfg3; do cp "$(ls -t gpurun_out/profiles_raw/trace_$c/*/*kernel_stats.csv | head -1)" profiles/r4/rocprofv3_kernel_stats_$c.csv; done
[ -f gpurun_out/parity_exceptions.md ] && cp gpurun_out/parity_exceptions.md profiles/r4/parity_exceptions.md
python - <<'PY'
import json
d = json.load(open('profiles/r4/bench_r4.json'))
print('headline', d['ms_per_step'], 'frac', d['roofline']['frac'], 'train', d['training']['ms_per_step'], 'cpu', d['cpu_baseline']['value'])
for k in ('bench_cfg2.json', 'bench_cfg3.json'):
    e = json.load(open('profiles/r4/' + k)); print(k, e['ms_per_step'], e['roofline']['frac'])
PY
cp "$(ls -t gpurun_out/profiles_raw/trace_compat/*/*kernel_stats.csv | head -1)" profiles/r4/rocprofv3_kernel_stats_cfg4_compat.csv
cp gpurun_out/profiles_raw/bench_compat.json profiles/r4/bench_cfg4_compat.json
