# A/B two prebuilt libraries on the training step: bash tools/ab_train.sh
for i in 1 2; do
for v in A B; do cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['training']['ms_per_step'])"; done; done
cp grafx_amd/lib/B.so grafx_amd/lib/libgrafx_amd.so
python -m pytest tests/test_gpu_autograd.py -q -m gpu 2>&1 | tail -1
