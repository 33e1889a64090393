# whole-step A/B of prebuilt libraries: bash tools/ab_bench.sh "A C"
for i in 1 2; do
for v in $1; do cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('render', d['ms_per_step'], 'frac', d['roofline']['frac'], 'train', d['training']['ms_per_step'])"; done; done
cp grafx_amd/lib/A.so grafx_amd/lib/libgrafx_amd.so
