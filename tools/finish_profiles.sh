#!/bin/bash
# after `gpurun -- bash tools/collect_profiles.sh` (which clears the raw directory first): distil the run into profiles/r5/
export GRAFX_ROUND=${GRAFX_ROUND:-r5}
D=profiles/$GRAFX_ROUND
mkdir -p $D
python tools/pmc_summary.py | tail -12
RAW=gpurun_out/profiles_raw
cp $RAW/bench_cfg2.json $D/bench_cfg2.json
cp $RAW/bench_cfg3.json $D/bench_cfg3.json
for c in cfg2 cfg3 longpole ballistics ballistics_rows; do
  f="$(ls -t $RAW/trace_$c/*/*kernel_stats.csv 2>/dev/null | head -1)"; [ -n "$f" ] && cp "$f" $D/rocprofv3_kernel_stats_$c.csv
done
f="$(ls -t $RAW/trace_compat/*/*kernel_stats.csv | head -1)"; cp "$f" $D/rocprofv3_kernel_stats_cfg4_compat.csv
cp $RAW/bench_compat.json $D/bench_cfg4_compat.json
cp $RAW/bench_longpole.json $D/bench_cfg4_longpole.json
cp $RAW/bench_ballistics.json $D/bench_cfg4_ballistics.json
cp $RAW/bench_longpole_r4path.json $D/bench_cfg4_longpole_r4path.json
cp $RAW/ballistics_bench.md $D/ballistics_bench.md
[ -f $RAW/alias_bench.md ] && cp $RAW/alias_bench.md $D/alias_bench.md
grep -v amdgpu.ids $RAW/mix_bench.txt > $D/mix_bench_longpole.txt
[ -f gpurun_out/parity_exceptions.md ] && cp gpurun_out/parity_exceptions.md $D/parity_exceptions.md
[ -f gpurun_out/measured_errors.json ] && cp gpurun_out/measured_errors.json $D/measured_errors.json
python - <<PY
import json
d = json.load(open('$D/bench_$GRAFX_ROUND.json'))
print('headline', d['ms_per_step'], 'frac', d['roofline']['frac'], 'train', d['training']['ms_per_step'], 'cpu', d['cpu_baseline']['value'])
for k, v in d['secondary'].items():
    print(k, v.get('ms_per_step'))
PY
