"""python tools/dyn_bwd_mix_bench.py: the compressor backward of the console graph (8192 rows = 256 graphs x 32 strips)
with its output gradient (a) stored rows written by the routing-sum adjoint first, (b) formed on the fly from the 5
destination gradients (gfx_dynamics_bwd_u1_mix_f32)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grafx_amd import ops  # noqa: E402

B, n, C, L, J, N = 256, 32, 2, 131072, 5, 16383
torch.manual_seed(0)
dev = torch.device("cuda")
x = torch.randn(B, n, C, L, device=dev)
gdst = torch.randn(B, J, C, L, device=dev)
gy = torch.empty(B, n, C, L, device=dev)
gx = torch.empty(B, n, C, L, device=dev)
R = B * n
p = [0.1 * torch.randn(R, 1, device=dev) for _ in range(4)]
u1 = torch.empty(R, L, device=dev)
ops.dynamics_fused(x, p[0], p[1], p[2], p[3], smoother=1, iir_len=N, knee="quadratic", gate=False, u1_out=u1)
dests = [[j // 8, 4] for j in range(n)]
lists = torch.tensor([sum((d + 1) << (16 * k) for k, d in enumerate(l)) for l in dests], device=dev)
slots = torch.arange(J, device=dev)
smask = torch.tensor([sum(1 << j for j in range(n) if d in dests[j]) for d in range(J)], device=dev)


def stored():
    assert ops.gather_sum_fanout(gdst, slots, smask, gy)
    return ops.dynamics_bwd(x, gy, p[0], p[1], p[2], p[3], N, "quadratic", False, out=gx, u1=u1)


def fused():
    rec = {"g": gdst, "lists": lists, "max_dests": 2}
    res = ops.dynamics_bwd(x, gy, p[0], p[1], p[2], p[3], N, "quadratic", False, out=gx, u1=u1, gmix=rec)
    assert res is not None and rec.get("done")
    return res


ref = [t.clone() for t in stored()]
got = fused()
print("gx equal:", torch.equal(ref[0], got[0]), " gparams max diff:", float((ref[1] - got[1]).abs().max()))
for name, fn in (("stored", stored), ("fused", fused), ("stored", stored), ("fused", fused)):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:7s} {a.elapsed_time(b) / 5:7.3f} ms")
