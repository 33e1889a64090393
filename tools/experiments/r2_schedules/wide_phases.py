"""Phase durations of the wide fftconv kernel from a -DGFX_W_STAMP build (100 MHz wall-clock stamps of every 64th workgroup)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from grafx_amd import ops
from grafx_amd._lib import lib

torch.manual_seed(0)
B, n, L, N = 256, 32, 131072, 4001
x4 = torch.randn(B, n, 2, L, device="cuda")
h = torch.randn(n, 1, N, device="cuda") / N ** 0.5
Hs = ops.fir_spectrum(h.reshape(-1, N))
y = torch.empty(B, n, 2, L, device="cuda")
SCHED = sys.argv[1] if len(sys.argv) > 1 else "wide"
if len(sys.argv) > 2 and sys.argv[2] == "alias":   # every row reads the same signal: window loads hit L2
    x4 = x4[:1, :1].expand(B, n, 2, L)
for _ in range(3):
    ops.fftconv(x4, Hs, N, 1, out=y, h_rows=n, schedule=SCHED)
torch.cuda.synchronize()
nblocks = B * n * 2 * 11
nrec = nblocks // 64
buf = np.zeros(nrec * 12, dtype=np.uint64)
raw = ctypes.CDLL(lib()._name)
raw.gfx_dbg_stamp_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert raw.gfx_dbg_stamp_read(buf.ctypes.data, buf.nbytes) == 0
s = buf.reshape(nrec, 12).astype(np.int64)
if SCHED == "wide":
    names = ["issue loads", "wait window", "forward 1", "issue H", "forward 2", "forward 3", "product", "inverse", "issue stores", "drain stores"]
else:   # -DGFX_T_STAMP build of the 256-thread tile kernel
    names = ["issue loads", "wait loads", "forward", "product", "inverse", "issue stores", "drain stores"]
s = s[:, : len(names) + 1]
d = np.diff(s, axis=1) / 100.0          # microseconds
ok = (d >= 0).all(axis=1) & (d.sum(axis=1) < 1000)
print(f"{ok.sum()} of {nrec} records; kernel span {(s[ok, -1].max() - s[ok, 0].min()) / 100.0:.1f} us")
for i, nm in enumerate(names):
    print(f"  {nm:14s} mean {d[ok, i].mean():6.2f} us   p10 {np.percentile(d[ok, i], 10):6.2f}  p50 {np.percentile(d[ok, i], 50):6.2f}  p90 {np.percentile(d[ok, i], 90):6.2f}")
print(f"  {'total':14s} mean {d[ok].sum(axis=1).mean():6.2f} us")
