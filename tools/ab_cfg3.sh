# A/B prebuilt libraries on BASELINE configs[2] (the 60001-tap reverb): bash tools/ab_cfg3.sh "A IF" [rounds]
for i in $(seq ${2:-2}); do
for v in $1; do cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python bench.py --config cfg3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 ms', d['ms_per_step'])"; done; done
cp grafx_amd/lib/A.so grafx_amd/lib/libgrafx_amd.so
