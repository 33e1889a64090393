# A/B prebuilt libraries on the training step: bash tools/ab_train.sh "A AB" [rounds]
for i in $(seq ${2:-2}); do
for v in $1; do cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['training']['ms_per_step'], 'fwd', d['ms_per_step'])"; done; done
cp grafx_amd/lib/A.so grafx_amd/lib/libgrafx_amd.so
