# run a python script against several prebuilt libraries: bash tools/ab_py.sh "A B" script.py "<grep>"
for v in $1; do [ -f grafx_amd/lib/$v.so ] || continue; cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python $2 2>&1 | grep -E "$3"; done
cp grafx_amd/lib/A.so grafx_amd/lib/libgrafx_amd.so
