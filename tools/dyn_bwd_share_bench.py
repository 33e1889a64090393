"""ops.dynamics_bwd at the console's channel-strip shape with the output gradient (a) one row per strip, (b) in block
form -- eight strips read one row through a zero stride (render/graph.py: _block_fan):
    GRAFX_DYN_BWD_SHARE=0|1 python tools/dyn_bwd_share_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grafx_amd import ops  # noqa: E402

B, n, m, L, N = 256, 32, 8, 131072, 16383
R = B * n
torch.manual_seed(0)
x = torch.randn(R, 2, L, device="cuda")
p = [0.1 * torch.randn(R, 1, device="cuda") for _ in range(4)]
u1 = torch.empty(R, L, device="cuda")
ops.dynamics_fused(x, p[0], p[1], p[2], p[3], smoother=1, iir_len=N, knee="quadratic", gate=False, u1_out=u1)
gx = torch.empty_like(x)
rows = torch.randn(R // m, 1, 2, L, device="cuda")
forms = {"block": rows.expand(-1, m, -1, -1), "expanded": rows.expand(-1, m, -1, -1).reshape(R, 2, L).contiguous()}
out = {}
for name, gy in forms.items():
    for rep in range(2):
        gxx, gp, da = ops.dynamics_bwd(x, gy, p[0], p[1], p[2], p[3], N, "quadratic", False, out=gx, u1=u1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            ops.dynamics_bwd(x, gy, p[0], p[1], p[2], p[3], N, "quadratic", False, out=gx, u1=u1)
        b.record()
        torch.cuda.synchronize()
        print(f"dynamics_bwd gy={name:9s} share={os.environ.get('GRAFX_DYN_BWD_SHARE', '1')} {a.elapsed_time(b) / 3:8.3f} ms")
    out[name] = (gx.clone(), gp.clone(), da.clone())
for a, b in zip(out["block"], out["expanded"]):
    assert torch.equal(a, b)
print("same bits")
