# bash tools/experiments/r4_tile16/run_half_variants.sh   (on the GPU box; code objects built beforehand, see gen_half_ablation.py)
echo "== shipped kernels (256 threads per tile)"
timeout 90 python tools/microbench.py eqbuf --rows 8192 --iters 5 2>&1 | grep -E "pipe|tile " | head -8
for v in half half_nostore half_noh half_noload half_noload_nostore_noh; do
  echo "== $v (512 threads per tile, timing only)"
  GRAFX_PIPE_HSACO=grafx_amd/lib/hs/$v.hsaco GRAFX_PIPE_THREADS=512 timeout 60 python tools/microbench.py eqbuf --rows 8192 --iters 5 2>&1 | grep -E "pipe|fault" | head -5
done
