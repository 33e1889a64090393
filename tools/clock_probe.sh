#!/bin/bash
# effective shader clock (GRBM_GUI_ACTIVE / kernel duration) of the fftconv kernels for several prebuilt libraries:
#   bash tools/clock_probe.sh "A W7 W8"      (libraries grafx_amd/lib/<NAME>.so; microbench eqbuf at 8192 rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/clock
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  export GRAFX_AMD_LIB=$R/grafx_amd/lib/$v.so   # the variant is selected by name, the live library is not touched
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$v -- python3 $R/tools/microbench.py eqbuf --rows 8192 --iters 3 > $OUT/$v.log 2>&1
done
unset GRAFX_AMD_LIB
python3 - <<PY
import csv, glob, collections, re, os
for v in "$1".split():
    dur = {}
    for f in glob.glob(f"$OUT/{v}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(list)
    for f in glob.glob(f"$OUT/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
                continue
            name, ns = dur[r["Dispatch_Id"]]
            if "fftconv1" in name and ns > 1e6:
                k = re.sub(r"\(.*$", "", re.sub(r"^void\s+", "", name))
                agg[k].append((float(r["Counter_Value"]) / ns, ns / 1e6))
    for k, xs in sorted(agg.items()):
        print(f"{v:6s} {k:45s} n={len(xs):2d} clock {sum(x[0] for x in xs)/len(xs):.3f} GHz  mean {sum(x[1] for x in xs)/len(xs):.3f} ms")
PY
