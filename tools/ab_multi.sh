# time several prebuilt libraries (grafx_amd/lib/<NAME>.so) on one box: bash tools/ab_multi.sh "A B C" "<microbench args>" "<grep>" [rounds]
for i in $(seq ${4:-2}); do
for v in $1; do [ -f grafx_amd/lib/$v.so ] || continue; cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python tools/microbench.py $2 2>&1 | grep -E "$3"; done; done
cp grafx_amd/lib/A.so grafx_amd/lib/libgrafx_amd.so
