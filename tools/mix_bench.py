"""python tools/mix_bench.py: the compressor stage + routing sum of the console graph (8192 rows, 32 per graph -> 4 buses +
send) as two kernels and as the fused one; HIP-event time per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafx_amd import ops

B, n, C, L, J = 256, 32, 2, 131072, 5
dev = torch.device("cuda")
torch.manual_seed(0)
buf = torch.empty(B, 2 * n + J, C, L, device=dev)
buf[:, :n].normal_()
x, y, mo = buf[:, :n], buf[:, n : 2 * n], buf[:, 2 * n :]
p = [torch.randn(n, 1, device=dev) * 0.1 for _ in range(4)]
if os.environ.get("MIX_BENCH_Z"):   # every smoother logit at this value (6 -> pole 0.9975: the long-memory rows)
    p[3] = torch.full_like(p[3], float(os.environ["MIX_BENCH_Z"]))
print(f"z_alpha = {os.environ.get('MIX_BENCH_Z', 'randn*0.1')}  lookback={ops.DYN_LOOKBACK}  "
      f"GRAFX_DYN_DEFER={os.environ.get('GRAFX_DYN_DEFER', 'auto')}  lib={os.environ.get('GRAFX_AMD_LIB', 'default')}")
dests = [list(range(8 * k, 8 * k + 8)) for k in range(4)] + [list(range(n))]
codes, n_acc, _, _ = ops.mix_schedule(dests, n)
sched = torch.tensor(codes, device=dev)
uniq = torch.arange(n, device=dev)
masks = torch.tensor([sum(1 << d for d, rows in enumerate(dests) if j in rows) for j in range(n)], device=dev)
kw = dict(smoother=1, iir_len=16383, knee="quadratic", gate=False, param_rows=n, out=y)


def separate():
    ops.dynamics_fused(x, *p, **kw)
    assert ops.gather_sum_fanout(y, uniq, masks, mo)


def fused():
    mix = {"sched": sched, "n_acc": n_acc, "out": mo}
    ops.dynamics_fused(x, *p, **kw, mix=mix)
    assert mix.get("done")


for name, fn in (("separate", separate), ("fused", fused), ("separate", separate), ("fused", fused)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (2 * B * n * C * L * 4 + B * J * C * L * 4) / 1e9
    print(f"{name:9s} {ms:7.3f} ms   ({gb / ms:6.2f} TB/s of the fused kernel's {gb:.1f} GB)")
