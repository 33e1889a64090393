#!/bin/bash
# kernel trace of the training step (headline forward reduced to 1 step): bash tools/trace_train.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_train_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --train-steps 3 --no-secondary --no-sustained > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms max {float(r['MaxNs'])/1e6:8.3f} tot {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
grep -o '"training": {[^}]*}' $OUT/bench.json | head -1
