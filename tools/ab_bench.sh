# per-kernel ms/step of bench.py for several prebuilt libraries: bash tools/ab_bench.sh "A B" [extra bench args]
for v in $1; do [ -f grafx_amd/lib/$v.so ] || continue; cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['roofline']['per_kernel_ms_per_step'])"; done
cp grafx_amd/lib/A.so grafx_amd/lib/libgrafx_amd.so
