"""time ops.dynamics_bwd (kept scan) with the row and the one-shot schedule at the console shape: python tools/dyn_bwd_bench.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grafx_amd import ops  # noqa: E402

R, L, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 131072, 16383
torch.manual_seed(0)
x = torch.randn(R, 2, L, device="cuda")
gy = torch.randn(R, 2, L, device="cuda")
p = [0.1 * torch.randn(R, 1, device="cuda") for _ in range(4)]
u1 = torch.empty(R, L, device="cuda")
y = ops.dynamics_fused(x, p[0], p[1], p[2], p[3], smoother=1, iir_len=N, knee="quadratic", gate=False, u1_out=u1)
gx = torch.empty_like(x)
for sched in ("rows", "oneshot", "rows", "oneshot"):
    ops.dynamics_bwd(x, gy, p[0], p[1], p[2], p[3], N, "quadratic", False, out=gx, u1=u1, schedule=sched)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        ops.dynamics_bwd(x, gy, p[0], p[1], p[2], p[3], N, "quadratic", False, out=gx, u1=u1, schedule=sched)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    print(f"dynamics_bwd {sched:8s} R={R} {ms:8.3f} ms  {28 * R * L / ms / 1e6:8.1f} GB/s (7 row-units)")
