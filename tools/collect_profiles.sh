#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): the bench line, the rocprofv3 kernel
# trace of the same command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE, each alone with --kernel-trace).
# Everything lands in gpurun_out/profiles_raw/; tools/pmc_summary.py (run in the repo afterwards) distils it
# into profiles/r4/ (GRAFX_ROUND).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_raw
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/trace_cfg2 $OUT/trace_cfg3   # one run per directory
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-secondary --no-sustained > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-secondary --no-sustained > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-secondary --no-sustained > /dev/null 2> $OUT/pmc_write.err
ls -R $OUT | head -40
# BASELINE configs[1] / configs[2] as their own bench lines + kernel traces
for cfg in cfg2 cfg3; do
  python3 $R/bench.py --config $cfg > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 $R/bench.py --config $cfg --no-cpu-baseline > /dev/null 2> $OUT/trace_$cfg.err
done
# the compat console (upstream's default tap counts, batch 64) as a kernel trace of its own
rm -rf $OUT/trace_compat
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_compat -- python3 $R/bench.py --reference-default-lengths --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-secondary --no-sustained > $OUT/bench_compat.json 2> $OUT/trace_compat.err
