"""python tools/power_probe.py: shader clock and board power (amdgpu hwmon, sampled every 50 ms) while ONE kernel of the
headline step runs back to back for ~3 s: the persistent convolution kernel with / without the source write-through, the
compiler-scheduled one-tile-per-workgroup kernel, the fused dynamics + routing-sum kernel, a plain copy, and an idle
baseline.  Tells how much of the gap between a kernel's vector-ALU time and its duration is paid in CLOCK (the board's
power cap) rather than in stalls.  -> markdown table on stdout (profiles/r4/power_probe.md)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from grafx_amd import ops

B, n, C, L, J, N = 256, 32, 2, 131072, 5, 4001
dev = torch.device("cuda")
torch.manual_seed(0)
x = torch.randn(B, n, C, L, device=dev)
buf = torch.empty(B, 3 * n + J, C, L, device=dev)
src, eq, comp, mo = buf[:, :n], buf[:, n:2 * n], buf[:, 2 * n:3 * n], buf[:, 3 * n:]
h = torch.randn(n, 1, N, device=dev) / N ** 0.5
Hs = ops.fir_spectrum(h.reshape(n, N))
p = [torch.randn(n, 1, device=dev) * 0.1 for _ in range(4)]
dests = [list(range(8 * k, 8 * k + 8)) for k in range(4)] + [list(range(n))]
codes, n_acc, _, _ = ops.mix_schedule(dests, n)
sched = torch.tensor(codes, device=dev)
kw = dict(smoother=1, iir_len=16383, knee="quadratic", gate=False, param_rows=n)
ops.fftconv(x, Hs, N, 1, out=eq, tee=src, h_rows=n)


def dyn():
    mix = {"sched": sched, "n_acc": n_acc, "out": mo}
    ops.dynamics_fused(eq, *p, **kw, out=comp, mix=mix)


cases = [
    ("idle (sleep)", None, 0),
    ("gfx_fftconv_pipe_t1_o8 (conv + source write-through)", lambda: ops.fftconv(x, Hs, N, 1, out=eq, tee=src, h_rows=n), 25.77e9),
    ("gfx_fftconv_pipe_t0_o8 (conv only)", lambda: ops.fftconv(x, Hs, N, 1, out=eq, h_rows=n), 17.18e9),
    ("fftconv1_kernel<false> (one tile per workgroup, hipcc)", lambda: ops.fftconv(x, Hs, N, 1, out=eq, h_rows=n, schedule="tile"), 17.18e9),
    ("dyn_oneshot_mix_kernel (compressor + routing sums)", dyn, 18.5e9),
    ("copy (torch, 8.6 GB)", lambda: src.copy_(x), 17.18e9),
]
print("| kernel (8192 stereo rows x 131072, back to back for ~3 s) | ms / launch | of 8 TB/s | sclk mean (min) MHz | board power W |")
print("|---|---|---|---|---|")
for name, fn, nbytes in cases:
    if fn is None:
        with bench.GpuSampler() as smp:
            time.sleep(2.0)
        s = smp.summary()
        print(f"| {name} | - | - | {s['sclk_mhz_mean']:.0f} ({s['sclk_mhz_min']:.0f}) | {s['power_w_mean']:.0f} |", flush=True)
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    reps = max(20, int(3000 / e0.elapsed_time(e1)))
    with bench.GpuSampler() as smp:
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    s = smp.summary()
    print(f"| {name} | {ms:.3f} | {nbytes / (ms * 1e-3) / 8e12:.3f} | {s['sclk_mhz_mean']:.0f} ({s['sclk_mhz_min']:.0f}) | "
          f"{s['power_w_mean']:.0f} |", flush=True)
