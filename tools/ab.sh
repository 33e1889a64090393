#!/bin/bash
# A/B several prebuilt libraries on one box without touching the live one:
#   bash tools/ab.sh "A B" 2 python tools/microbench.py eq --rows 8192
# runs the command once per variant and round with GRAFX_AMD_LIB=grafx_amd/lib/<NAME>.so (grafx_amd/build.py honours it;
# build variants with tools/build_variant.sh).  stderr goes to gpurun_out/ab_<NAME>.err.
R=${GRAFT_REPO_ROOT:-$(pwd)}
variants=$1; rounds=$2; shift 2
mkdir -p $R/gpurun_out
for i in $(seq $rounds); do
  for v in $variants; do
    lib=$R/grafx_amd/lib/$v.so
    [ -f $lib ] || { echo "== $v: $lib missing, skipped"; continue; }
    echo "== $v (round $i)"
    GRAFX_AMD_LIB=$lib "$@" 2>>$R/gpurun_out/ab_$v.err || echo "   command failed for $v (see gpurun_out/ab_$v.err)"
  done
done
