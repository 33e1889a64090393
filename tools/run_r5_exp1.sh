python -m pytest tests/test_gpu_ballistics.py tests/test_gpu_dyn_lookback.py tests/test_gpu_mix_fusion.py tests/test_gpu_processors.py -q -m gpu 2>&1 | tail -40
python tools/ballistics_bench.py --rows 9216 256 2>&1 | tee gpurun_out/ballistics_bench.md | tail -30
echo "== mix bench"
MIX_BENCH_Z=6 GRAFX_DYN_LOOKBACK=0 python tools/mix_bench.py 2>&1 | grep -v amdgpu.ids
MIX_BENCH_Z=6 python tools/mix_bench.py 2>&1 | grep -v amdgpu.ids
python tools/mix_bench.py 2>&1 | grep -v amdgpu.ids
python bench.py > gpurun_out/bench_r5b.json 2> gpurun_out/bench_r5b.err; tail -c 300 gpurun_out/bench_r5b.err
