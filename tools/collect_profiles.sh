#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): the bench line, the rocprofv3 kernel
# trace of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each alone with --kernel-trace).
# Everything lands in gpurun_out/profiles_raw/; tools/finish_profiles.sh (run in the repo afterwards) distils it
# into profiles/r6/ (GRAFX_ROUND).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-train --no-secondary --no-sustained"
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 $LEAN > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $LEAN > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 $LEAN > /dev/null 2> $OUT/pmc_write.err
# round 6: the TRAINING step (forward + backward + flat all-reduce at 256 graphs): kernel trace and the two PMC passes
TRAIN="--steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-sustained"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -- python3 $R/bench.py $TRAIN --train-steps 3 > $OUT/bench_train.json 2> $OUT/trace_train.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_train -- python3 $R/bench.py $TRAIN --train-steps 1 > /dev/null 2> $OUT/pmc_fetch_train.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_train -- python3 $R/bench.py $TRAIN --train-steps 1 > /dev/null 2> $OUT/pmc_write_train.err
python3 $R/tools/step_timeline.py $OUT/trace_train > $OUT/train_step_timeline.txt 2>&1
# BASELINE configs[1] / configs[2] as their own bench lines + kernel traces
for cfg in cfg2 cfg3; do
  python3 $R/bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline > /dev/null 2> $OUT/trace_$cfg.err
done
# the compat console (upstream's default tap counts) at the stated batch 256, as a kernel trace of its own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_compat -- python3 $R/bench.py --reference-default-lengths --steps 2 --warmup 1 $LEAN > $OUT/bench_compat.json 2> $OUT/trace_compat.err
# the console with long compressor poles, at the clamp and with the ballistics smoother -- kernel traces (+ PMC for two)
for v in longpole clamp ballistics; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/bench.py --console-variant $v --steps 10 --warmup 3 $LEAN > $OUT/bench_$v.json 2> $OUT/trace_$v.err
done
for v in longpole ballistics; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$v -- python3 $R/bench.py --console-variant $v --steps 2 --warmup 1 $LEAN > /dev/null 2> $OUT/pmc_fetch_$v.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$v -- python3 $R/bench.py --console-variant $v --steps 2 --warmup 1 $LEAN > /dev/null 2> $OUT/pmc_write_$v.err
done
# the ballistics recursion on its own (9216 x 131072 rows, both coefficient regimes, both schedules)
python3 $R/tools/ballistics_bench.py > $OUT/ballistics_bench.md 2> $OUT/ballistics_bench.err
# the compressor stage + routing sum alone, fast and long poles
( python3 $R/tools/mix_bench.py; MIX_BENCH_Z=6 python3 $R/tools/mix_bench.py ) > $OUT/mix_bench.txt 2>&1
# the odd-length aliasing alone: one row against two rows per chirp-z transform, with and without the per-row scaling
python3 $R/tools/alias_bench.py 2> /dev/null | grep "^|" > $OUT/alias_bench.md
python3 $R/tools/alias_bench.py --rows 1024 --P 483999 299999 2> /dev/null | grep "^| [0-9]" >> $OUT/alias_bench.md
echo "without the per-row scaling of a pair (GRAFX_ALIAS_PAIR_SCALE=0):" >> $OUT/alias_bench.md
GRAFX_ALIAS_PAIR_SCALE=0 python3 $R/tools/alias_bench.py 2> /dev/null | grep "^| [0-9]" >> $OUT/alias_bench.md
GRAFX_ALIAS_PAIR_SCALE=0 python3 $R/tools/alias_bench.py --rows 1024 --P 483999 299999 2> /dev/null | grep "^| [0-9]" >> $OUT/alias_bench.md
# the compressor backward with its output gradient per row / in block form (training)
python3 $R/tools/dyn_bwd_share_bench.py 2> /dev/null | grep -v amdgpu > $OUT/dyn_bwd_block_bench.txt
ls $OUT | head -80
