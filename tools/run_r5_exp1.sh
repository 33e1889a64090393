GRAFX_DYN_DEFER=1 python -m pytest tests/test_gpu_dyn_lookback.py tests/test_gpu_mix_fusion.py tests/test_gpu_render.py -q -m gpu --tb=short 2>&1 | tail -8
python -m pytest tests/test_gpu_dyn_lookback.py tests/test_gpu_mix_fusion.py -q -m gpu --tb=short 2>&1 | tail -4
echo "== mix bench"
MIX_BENCH_Z=6 python tools/mix_bench.py 2>&1 | grep -v amdgpu.ids
GRAFX_DYN_DEFER=1 python tools/mix_bench.py 2>&1 | grep -v amdgpu.ids
python tools/mix_bench.py 2>&1 | grep -v amdgpu.ids
