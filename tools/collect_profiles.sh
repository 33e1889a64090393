#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): the bench line, the rocprofv3 kernel
# trace of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each alone with --kernel-trace).
# Everything lands in gpurun_out/profiles_raw/; tools/finish_profiles.sh (run in the repo afterwards) distils it
# into profiles/r5/ (GRAFX_ROUND).
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
# BASELINE configs[1] / configs[2] as their own bench lines + kernel traces
for cfg in cfg2 cfg3; do
  python3 $R/bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline > /dev/null 2> $OUT/trace_$cfg.err
done
# the compat console (upstream's default tap counts, batch 64) as a kernel trace of its own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_compat -- python3 $R/bench.py --reference-default-lengths --batch 64 --steps 5 --warmup 2 $LEAN > $OUT/bench_compat.json 2> $OUT/trace_compat.err
# round 5: the console with long compressor poles and with the ballistics smoother -- kernel traces and PMC passes
for v in longpole ballistics; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/bench.py --console-variant $v --steps 10 --warmup 3 $LEAN > $OUT/bench_$v.json 2> $OUT/trace_$v.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$v -- python3 $R/bench.py --console-variant $v --steps 2 --warmup 1 $LEAN > /dev/null 2> $OUT/pmc_fetch_$v.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$v -- python3 $R/bench.py --console-variant $v --steps 2 --warmup 1 $LEAN > /dev/null 2> $OUT/pmc_write_$v.err
done
# the long-pole console on round 4's path (row kernel + read-back) for the before / after of DESIGN section 4.5
GRAFX_DYN_LOOKBACK=0 python3 $R/bench.py --console-variant longpole --steps 10 --warmup 3 $LEAN > $OUT/bench_longpole_r4path.json 2> $OUT/bench_longpole_r4path.err
# the ballistics recursion on its own (9216 x 131072 rows, both coefficient regimes, both schedules)
python3 $R/tools/ballistics_bench.py --old-lib $R/grafx_amd/lib/r4base.so > $OUT/ballistics_bench.md 2> $OUT/ballistics_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ballistics_rows -- python3 $R/tools/ballistics_bench.py --rows 9216 --iters 3 > /dev/null 2> $OUT/trace_ballistics_rows.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_ballistics_rows -- python3 $R/tools/ballistics_bench.py --rows 9216 --iters 1 > /dev/null 2> $OUT/pmc_fetch_ballistics_rows.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_ballistics_rows -- python3 $R/tools/ballistics_bench.py --rows 9216 --iters 1 > /dev/null 2> $OUT/pmc_write_ballistics_rows.err
# the compressor stage + routing sum alone, fast and long poles
( python3 $R/tools/mix_bench.py; MIX_BENCH_Z=6 python3 $R/tools/mix_bench.py; MIX_BENCH_Z=6 GRAFX_DYN_LOOKBACK=0 python3 $R/tools/mix_bench.py ) > $OUT/mix_bench.txt 2>&1
ls $OUT | head -60
# the odd-length aliasing alone: one row against two rows per chirp-z transform
python3 $R/tools/alias_bench.py 2> /dev/null | grep "^|" > $OUT/alias_bench.md
python3 $R/tools/alias_bench.py --rows 1024 --P 483999 299999 2> /dev/null | grep "^| [0-9]" >> $OUT/alias_bench.md
