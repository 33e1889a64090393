#!/bin/bash
# after `gpurun -- bash tools/collect_profiles.sh` (which clears the raw directory first): distil the run into profiles/r6/
export GRAFX_ROUND=${GRAFX_ROUND:-r6}
D=profiles/$GRAFX_ROUND
mkdir -p $D
python tools/pmc_summary.py | tail -12
RAW=gpurun_out/profiles_raw
cp $RAW/bench_cfg2.json $D/bench_cfg2.json
cp $RAW/bench_cfg3.json $D/bench_cfg3.json
for c in cfg2 cfg3 longpole clamp ballistics train; do
  f="$(ls -t $RAW/trace_$c/*/*kernel_stats.csv 2>/dev/null | head -1)"; [ -n "$f" ] && cp "$f" $D/rocprofv3_kernel_stats_$c.csv
done
f="$(ls -t $RAW/trace_compat/*/*kernel_stats.csv | head -1)"; cp "$f" $D/rocprofv3_kernel_stats_cfg4_compat.csv
cp $RAW/bench_compat.json $D/bench_cfg4_compat.json
cp $RAW/bench_train.json $D/bench_train_under_rocprof.json
cp $RAW/train_step_timeline.txt $D/train_step_timeline.txt
for v in longpole clamp ballistics; do cp $RAW/bench_$v.json $D/bench_cfg4_$v.json; done
cp $RAW/ballistics_bench.md $D/ballistics_bench.md
[ -f $RAW/alias_bench.md ] && cp $RAW/alias_bench.md $D/alias_bench.md
[ -f $RAW/dyn_bwd_block_bench.txt ] && cp $RAW/dyn_bwd_block_bench.txt $D/dyn_bwd_block_bench.txt
grep -v amdgpu.ids $RAW/mix_bench.txt > $D/mix_bench_longpole.txt
[ -f gpurun_out/parity_exceptions.md ] && cp gpurun_out/parity_exceptions.md $D/parity_exceptions.md
[ -f gpurun_out/measured_errors.json ] && cp gpurun_out/measured_errors.json $D/measured_errors.json
python - <<PY
import json
d = json.load(open('$D/bench_$GRAFX_ROUND.json'))
print('headline', d['ms_per_step'], 'frac', d['roofline']['frac'], 'train', d['training']['ms_per_step'], 'cpu', d['cpu_baseline']['value'])
print(d['summary'])
PY
