"""Time gfx_iir_fsm_fir_f32 at the headline size (8192 filter rows x 6 biquads -> 4001 taps) and print a checksum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from grafx_amd import ops

torch.manual_seed(0)
for RC, K, N in [(8192, 6, 4001), (8192, 6, 4000), (8192, 1, 4001), (512, 6, 16384)]:
    Bs = torch.randn(RC, K, 3, device="cuda") * 0.3
    As = torch.randn(RC, K, 3, device="cuda") * 0.1
    As[..., 0] = 1.0
    plan = ops.iir_fsm_plan(N, Bs.device)
    h = ops.iir_fsm_fir(Bs, As, N, plan)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        h = ops.iir_fsm_fir(Bs, As, N, plan)
    e1.record()
    torch.cuda.synchronize()
    print(f"RC={RC} K={K} N={N}: {e0.elapsed_time(e1) / 10:.3f} ms   checksum {h.double().sum().item():.9e} {h.double().abs().sum().item():.9e}")
