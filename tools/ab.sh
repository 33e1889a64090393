# A/B two prebuilt libraries (grafx_amd/lib/A.so, B.so) on one box: bash tools/ab.sh
cp grafx_amd/lib/B.so grafx_amd/lib/libgrafx_amd.so
python -m pytest tests/test_gpu_fftconv.py tests/test_gpu_kernels_basic.py -q -m gpu -x 2>&1 | tail -2
for i in 1 2; do
for v in A B; do cp grafx_amd/lib/$v.so grafx_amd/lib/libgrafx_amd.so; echo "== $v"; python tools/microbench.py eq --rows 8192 2>&1 | grep -E "fftconv1"; python tools/microbench.py eq --rows 2048 2>&1 | grep -E "fftconv1 only"; done; done
cp grafx_amd/lib/B.so grafx_amd/lib/libgrafx_amd.so
