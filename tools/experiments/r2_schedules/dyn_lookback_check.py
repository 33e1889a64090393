import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafx_amd import ops
torch.manual_seed(0)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
# correctness across poles incl. the clamp and truncation-relevant ones
R, L = 6, 131072
x = torch.randn(R, 2, L, device='cuda') * torch.linspace(0.05, 1.0, L, device='cuda')
p = dict(log_threshold=torch.randn(R, 1, device='cuda') - 2, log_ratio=torch.randn(R, 1, device='cuda'),
         log_knee=torch.randn(R, 1, device='cuda'), z_alpha=torch.tensor([[20.0], [9.0], [6.0], [2.0], [0.0], [-3.0]], device='cuda'))
for iir_len in (16383, 1023, 300):
    for knee, gate in (("quadratic", False), ("hard", True), ("exponential", False)):
        a = ops.dynamics_fused(x.repeat(400, 1, 1), *(p[k].repeat(400, 1) for k in ("log_threshold", "log_ratio", "log_knee", "z_alpha")), smoother=1, iir_len=iir_len, knee=knee, gate=gate, schedule="rows")[:R]
        b = ops.dynamics_fused(x, p["log_threshold"], p["log_ratio"], p["log_knee"], p["z_alpha"], smoother=1, iir_len=iir_len, knee=knee, gate=gate, schedule="lookback")
        err = ((a - b).abs().amax(dim=(1, 2)) / a.abs().amax(dim=(1, 2))).tolist()
        print(iir_len, knee, gate, ["%.1e" % e for e in err], bool(torch.isfinite(b).all()))
# ragged length / mono
xr = torch.randn(3, 1, 5001, device='cuda')
pr = {k: v[:3] for k, v in p.items()}
a = ops.dynamics_fused(xr, pr["log_threshold"], pr["log_ratio"], pr["log_knee"], pr["z_alpha"], smoother=1, iir_len=63, knee="quadratic", gate=False, schedule="rows")
b = ops.dynamics_fused(xr, pr["log_threshold"], pr["log_ratio"], pr["log_knee"], pr["z_alpha"], smoother=1, iir_len=63, knee="quadratic", gate=False, schedule="lookback")
print("ragged", float((a - b).abs().max() / a.abs().max()))
# speed at the headline shape
R = 8192
X = torch.randn(R, 2, L, device='cuda'); Y = torch.empty_like(X)
P = {k: 0.1 * torch.randn(R, 1, device='cuda') for k in ("log_threshold", "log_ratio", "log_knee", "z_alpha")}
for sched in ("rows", "lookback", "rows", "lookback"):
    ms = timeit(lambda: ops.dynamics_fused(X, P["log_threshold"], P["log_ratio"], P["log_knee"], P["z_alpha"], smoother=1, iir_len=16383, knee="quadratic", gate=False, out=Y, schedule=sched))
    print(sched, "%.3f ms  %.0f GB/s" % (ms, 16 * R * L / ms / 1e6))
# slow poles at scale: a = clamp (z=20): long chains
P2 = dict(P); P2["z_alpha"] = torch.full((R, 1), 20.0, device='cuda')
for sched in ("rows", "lookback"):
    ms = timeit(lambda: ops.dynamics_fused(X, P2["log_threshold"], P2["log_ratio"], P2["log_knee"], P2["z_alpha"], smoother=1, iir_len=16383, knee="quadratic", gate=False, out=Y, schedule=sched))
    print("clamped pole", sched, "%.3f ms  %.0f GB/s" % (ms, 16 * R * L / ms / 1e6))
