#!/bin/bash
# quick kernel trace of the headline bench (3 steps): bash tools/trace_quick.sh <tag>; prints the top kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms max {float(r['MaxNs'])/1e6:8.3f} tot {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/bench.json | head -1
