#!/bin/bash
# time several builds of the hand-scheduled kernel (code objects under grafx_amd/lib/hs/, made with
# `python -m grafx_amd.csrc.asm.gen_fftconv_pipe --hsaco grafx_amd/lib/hs/NAME.hsaco knob=value ...`) on one box:
#   bash tools/pipe_variants.sh "base early late" [rows]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in $1; do
  f=$R/grafx_amd/lib/hs/$v.hsaco
  [ -f $f ] || { echo "== $v: missing"; continue; }
  echo "== $v"
  GRAFX_PIPE_HSACO=$f timeout 120 python $R/tools/microbench.py eqbuf --rows ${2:-8192} --iters 5 2>&1 | grep -E "pipe|tile .*(buffer out \+ tee|contiguous)"
done
