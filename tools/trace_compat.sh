#!/bin/bash
# kernel trace of the console graph with upstream's default (even) tap counts at batch 64: bash tools/trace_compat.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_compat
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --reference-default-lengths --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-secondary --no-sustained > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e6:8.3f} ms tot {float(r['TotalDurationNs'])/1e6:9.2f} ms {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
python3 -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print('ms/step', d['ms_per_step'])"
