"""A/B of the persistent fftconv1 experiment (GFX_PERSISTENT) inside one process."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grafx_amd import ops
import grafx_amd.processors as P

dev = "cuda"
torch.manual_seed(0)
L, n = 131072, 32
for R in (2048, 8192):
    B = R // n
    x4 = torch.randn(B, n, 2, L, device=dev)
    buf = torch.empty(B, 111, 2, L, device=dev)
    eq = P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=4001).to(dev)
    p = {k: 0.1 * torch.randn(n, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
    Bs, As = ops.peq_coeffs(p["w0"], p["q_inv"], p["log_gain"])
    Hs = ops.fir_spectrum(ops.iir_fsm_fir(Bs, As, 4001, eq.biquad._plan(x4.device)).reshape(n, 4001))
    ref = {}
    for rep in range(2):
        for mode in ("base", "persistent"):
            os.environ.pop("GFX_PERSISTENT", None)
            if mode == "persistent":
                os.environ["GFX_PERSISTENT"] = "1"
            for tee in (False, True):
                out, tcopy = buf.narrow(1, 32, n), (buf.narrow(1, 0, n) if tee else None)
                f = lambda: ops.fftconv(x4, Hs, 4001, 1, out=out, tee=tcopy, h_rows=n)
                f(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    f()
                e1.record(); torch.cuda.synchronize()
                key = (tee,)
                if mode == "base":
                    ref[key] = out.clone()
                    same = ""
                else:
                    same = " bit-identical" if torch.equal(out, ref[key]) else f" MAXDIFF {(out - ref[key]).abs().max().item():.3e}"
                    if tee:
                        same += " tee-ok" if torch.equal(buf.narrow(1, 0, n), x4) else " TEE-MISMATCH"
                print(f"R={R} {mode:10s} tee={tee!s:5s} {e0.elapsed_time(e1) / 5:7.3f} ms{same}")
    del x4, buf
