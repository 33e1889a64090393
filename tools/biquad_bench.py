"""Time of the exact recursive biquad cascade (gfx_biquad_cascade_f32, csrc/biquad.hip) at headline-like sizes:
   python tools/biquad_bench.py [--rows 4096] [--length 131072] [--sections 6] [--iters 10]
-> ms per call and GB/s over the algorithmic 8 bytes per channel-sample."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grafx_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4096)
ap.add_argument("--length", type=int, default=131072)
ap.add_argument("--sections", type=int, nargs="+", default=[1, 6])
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
torch.manual_seed(0)
x = torch.randn(a.rows, 2, a.length, device="cuda")
for K in a.sections:
    # stable sections: poles at radius 0.5-0.95
    r = 0.5 + 0.45 * torch.rand(a.rows, 2, K, device="cuda")
    th = 3.0 * torch.rand(a.rows, 2, K, device="cuda")
    As = torch.stack([torch.ones_like(r), -2 * r * torch.cos(th), r * r], -1)
    Bs = torch.randn(a.rows, 2, K, 3, device="cuda")
    out = torch.empty_like(x)
    for _ in range(3):
        ops.biquad_cascade(x, Bs, As, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        ops.biquad_cascade(x, Bs, As, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"K={K}: {ms:.3f} ms, {8 * x.numel() / ms / 1e6:.0f} GB/s")
