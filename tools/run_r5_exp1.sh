python -m pytest tests/test_gpu_ballistics.py -q -m gpu --tb=short 2>&1 | tail -4
python tools/ballistics_bench.py --rows 9216 256 2>&1 | grep -v amdgpu.ids
