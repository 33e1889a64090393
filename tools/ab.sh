# A/B two prebuilt libraries (grafx_amd/lib/A.so, B.so) on one box: bash tools/ab.sh "<microbench args>" "<grep pattern>"
for i in 1 2; do
for v in A B; do cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python tools/microbench.py $1 2>&1 | grep -E "$2"; done; done
cp grafx_amd/lib/B.so grafx_amd/lib/libgrafx_amd.so
