#!/usr/bin/env python3
"""Per-processor micro-benchmarks at headline shapes (run on the GPU box).

    python tools/microbench.py [eq|comp|reverb|mix|all] [--rows 2048] [--length 131072] [--iters 5]
Prints achieved algorithmic GB/s (4*(C_in+C_out)*R*L bytes per call) per processor call.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--length", type=int, default=131072)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    import grafx_amd.processors as P
    from grafx_amd import ops

    R, L = a.rows, a.length
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(R, 2, L, device=dev)
    y = torch.empty_like(x)
    gb = 16 * R * L / 1e9
    with torch.no_grad():
        if a.what in ("eq", "all"):
            eq = P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=4001).to(dev)
            p = {k: 0.1 * torch.randn(R, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
            ms = timeit(lambda: eq.render_into(x.view(1, R, 2, L), y.view(1, R, 2, L), **p), a.iters)
            print(f"eq      R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            Bs, As = ops.peq_coeffs(p["w0"], p["q_inv"], p["log_gain"])
            h = ops.iir_fsm_fir(Bs, As, 4001, eq.biquad._plan(x.device))
            Hs = ops.fir_spectrum(h)
            ms = timeit(lambda: ops.fftconv(x, Hs, 4001, 1, out=y), a.iters)
            print(f"  fftconv1 only       {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            tee = torch.empty_like(x)
            ms = timeit(lambda: ops.fftconv(x, Hs, 4001, 1, out=y, tee=tee), a.iters)
            print(f"  fftconv1 + tee      {ms:8.3f} ms  {1.5 * gb / ms * 1e3:8.1f} GB/s")
            ms = timeit(lambda: tee.copy_(x), a.iters)
            print(f"  plain copy          {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            for sched in ("tile", "pipe"):
                ms = timeit(lambda: ops.fftconv(x, Hs, 4001, 1, out=y, schedule=sched), a.iters)
                print(f"  fftconv1 {sched:9s}      {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
                ms = timeit(lambda: ops.fftconv(x, Hs, 4001, 1, out=y, tee=tee, schedule=sched), a.iters)
                print(f"  fftconv1 {sched:9s} +tee {ms:8.3f} ms  {1.5 * gb / ms * 1e3:8.1f} GB/s")
            ms = timeit(lambda: ops.iir_fsm_fir(Bs, As, 4001, eq.biquad._plan(x.device)), a.iters)
            print(f"  iir_fsm only        {ms:8.3f} ms")
            ms = timeit(lambda: ops.fir_spectrum(h), a.iters)
            print(f"  hspec only          {ms:8.3f} ms")
        if a.what in ("eqbuf",):   # the first stage exactly as the console render runs it: strided views of the buffer
            B, n, V = R // 32, 32, 111
            eq = P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=4001).to(dev)
            p = {k: 0.1 * torch.randn(n, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
            Bs, As = ops.peq_coeffs(p["w0"], p["q_inv"], p["log_gain"])
            Hs = ops.fir_spectrum(ops.iir_fsm_fir(Bs, As, 4001, eq.biquad._plan(x.device)).reshape(n, 4001))
            x4 = x.view(B, n, 2, L)
            buf = torch.empty(B, V, 2, L, device=dev)
            for name, out, tee, xin in (("contiguous in/out, shared H", y.view(B, n, 2, L), None, x4),
                                        ("buffer out", buf.narrow(1, 32, n), None, x4),
                                        ("buffer out + tee", buf.narrow(1, 32, n), buf.narrow(1, 0, n), x4),
                                        ("buffer in/out", buf.narrow(1, 64, n), None, buf.narrow(1, 32, n))):
                for sched in ("tile", "pipe"):
                    ms = timeit(lambda: ops.fftconv(xin, Hs, 4001, 1, out=out, tee=tee, h_rows=n, schedule=sched), a.iters)
                    print(f"  fftconv1 {sched:5s} {name:32s} {ms:8.3f} ms  {(1.5 if tee is not None else 1.0) * gb / ms * 1e3:8.1f} GB/s")
        if a.what in ("eqcold",):  # the tee launch back to back vs. after other traffic (as inside a render step)
            B, n, V = R // 32, 32, 111
            eq = P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=4001).to(dev)
            p = {k: 0.1 * torch.randn(n, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
            Bs, As = ops.peq_coeffs(p["w0"], p["q_inv"], p["log_gain"])
            Hs = ops.fir_spectrum(ops.iir_fsm_fir(Bs, As, 4001, eq.biquad._plan(x.device)).reshape(n, 4001))
            x4 = x.view(B, n, 2, L)
            bufs = [torch.empty(B, V, 2, L, device=dev) for _ in range(2)]
            other = torch.empty(B, 8, 2, L, device=dev)

            def timed(prep, k):
                ts = []
                for i in range(a.iters + 1):
                    buf = bufs[i % k]
                    prep(buf)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.fftconv(x4, Hs, 4001, 1, out=buf.narrow(1, 32, n), tee=buf.narrow(1, 0, n), h_rows=n)
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                return sum(ts[1:]) / a.iters

            for reps in (1, 5, 20, 60):  # sustained load: does the per-launch time creep up (clocks)?
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(reps):
                    ops.fftconv(x4, Hs, 4001, 1, out=bufs[0].narrow(1, 32, n), tee=bufs[0].narrow(1, 0, n), h_rows=n)
                e1.record()
                torch.cuda.synchronize()
                print(f"  {reps:3d} tee launches in a row: {e0.elapsed_time(e1) / reps:8.3f} ms each")
            cp = P.Compressor(energy_smoother="iir", iir_len=16383, flashfftconv=False).to(dev)
            pc = {k: 0.1 * torch.randn(n, 1, device=dev) for k in cp.parameter_size()}
            ts = []
            for i in range(8):  # interleaved with the compressor stage, as in the render
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.fftconv(x4, Hs, 4001, 1, out=bufs[0].narrow(1, 32, n), tee=bufs[0].narrow(1, 0, n), h_rows=n)
                e1.record()
                cp.render_into(bufs[0].narrow(1, 32, n), bufs[0].narrow(1, 64, n), _shared_rows=n, **pc)
                cp.render_into(bufs[0].narrow(1, 32, n), bufs[0].narrow(1, 64, n), _shared_rows=n, **pc)
                ts.append((e0, e1))
            torch.cuda.synchronize()
            print("  tee launch interleaved with 2 compressor stages: " + " ".join(f"{a_.elapsed_time(b_):.2f}" for a_, b_ in ts))
            def design():
                b_, a_ = ops.peq_coeffs(p["w0"], p["q_inv"], p["log_gain"])
                return ops.fir_spectrum(ops.iir_fsm_fir(b_, a_, 4001, eq.biquad._plan(x.device)).reshape(n, 4001))

            for label, pre in (("2 compressor stages, then the 3 design kernels", True), ("2 compressor stages only", False)):
                ts = []
                for i in range(8):
                    cp.render_into(bufs[0].narrow(1, 32, n), bufs[0].narrow(1, 64, n), _shared_rows=n, **pc)
                    cp.render_into(bufs[0].narrow(1, 32, n), bufs[0].narrow(1, 64, n), _shared_rows=n, **pc)
                    H2 = design() if pre else Hs
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.fftconv(x4, H2, 4001, 1, out=bufs[0].narrow(1, 32, n), tee=bufs[0].narrow(1, 0, n), h_rows=n)
                    e1.record()
                    ts.append((e0, e1))
                torch.cuda.synchronize()
                print(f"  tee launch after {label}: " + " ".join(f"{a_.elapsed_time(b_):.2f}" for a_, b_ in ts))
            print(f"  tee launch, back to back, one buffer      {timed(lambda b: None, 1):8.3f} ms")
            print(f"  tee launch, alternating two buffers       {timed(lambda b: None, 2):8.3f} ms")
            print(f"  tee launch after rewriting the other 30 GB buffer {timed(lambda b: bufs[1].fill_(1.0), 1):8.3f} ms")
            print(f"  tee launch after rewriting its own 30 GB buffer   {timed(lambda b: b.fill_(1.0), 1):8.3f} ms")
            print(f"  tee launch after reading x and the buffer (sum)   {timed(lambda b: (b.sum(), x4.sum()), 1):8.3f} ms")
            print(f"  tee launch after a 2 GB copy              {timed(lambda b: other.copy_(b.narrow(1, 100, 8)), 1):8.3f} ms")
            print(f"  tee launch after writing the buffer tail  {timed(lambda b: b.narrow(1, 64, 47).fill_(1.0), 2):8.3f} ms")
        if a.what in ("grad",):  # the equaliser's filter gradient: one-pass correlation vs partitioned form
            g = torch.randn_like(x)
            ms = timeit(lambda: ops.fir_grad(x, g, 4001, 0), a.iters)
            print(f"fir_grad (corr1)            R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            Pl = ops.part_len_for(L, 4001)
            ms = timeit(lambda: ops.fftconv(g, ops.fir_spectrum_reversed(x, part_len=Pl), L, 2, Lout=4001, off=L - 1, part_len=Pl), a.iters)
            print(f"hspec<rev> + winmac         R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
        if a.what in ("eqx", "all"):
            eqx = P.ParametricEqualizer(num_filters=6, backend="lfilter", flashfftconv=False).to(dev)
            p = {k: 0.1 * torch.randn(R, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
            ms = timeit(lambda: eqx(x, **p), a.iters)
            print(f"eq exact recursion (K=6) R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            Bs, As = ops.peq_coeffs(p["w0"], p["q_inv"], p["log_gain"])
            for K in (1, 2, 6):
                ms = timeit(lambda: ops.biquad_cascade(x, Bs[:, :, :K].contiguous(), As[:, :, :K].contiguous(), out=y), a.iters)
                print(f"  biquad_cascade K={K}   {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
        if a.what in ("comp", "all"):
            cp = P.Compressor(energy_smoother="iir", iir_len=16383, flashfftconv=False).to(dev)
            p = {k: 0.1 * torch.randn(R, 1, device=dev) for k in cp.parameter_size()}
            for label, z in (("poles sigmoid(0.1 randn)", p["z_alpha_pre"]), ("poles 0.88 (H = 218)", torch.full((R, 1), 2.0, device=dev)),
                             ("poles 0.95 (row kernel)", torch.full((R, 1), 3.0, device=dev)),
                             ("poles at the clamp", torch.full((R, 1), 20.0, device=dev)),
                             ("half fast, half at the clamp", torch.where(torch.arange(R, device=dev)[:, None] % 2 == 0, 0.0, 20.0))):
                for sched in ("rows", "oneshot"):
                    ops.DYN_SCHEDULE = sched
                    q = dict(p, z_alpha_pre=z)
                    ms = timeit(lambda: cp.render_into(x.view(1, R, 2, L), y.view(1, R, 2, L), **q), a.iters)
                    print(f"comp {sched:8s} {label:30s} R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            ops.DYN_SCHEDULE = "oneshot"
        if a.what in ("comp1",):   # one-shot vs row schedule at the headline poles, plus the copy the same bytes make
            cp = P.Compressor(energy_smoother="iir", iir_len=16383, flashfftconv=False).to(dev)
            p = {k: 0.1 * torch.randn(R, 1, device=dev) for k in cp.parameter_size()}
            for sched in ("rows", "oneshot", "rows", "oneshot"):
                ops.DYN_SCHEDULE = sched
                ms = timeit(lambda: cp.render_into(x.view(1, R, 2, L), y.view(1, R, 2, L), **p), a.iters)
                print(f"comp {sched:8s} R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
            ms = timeit(lambda: y.copy_(x), a.iters)
            print(f"torch copy    R={R} {ms:8.3f} ms  {gb / ms * 1e3:8.1f} GB/s")
        if a.what in ("reverb", "all"):
            Rr = max(R // 8, 1)
            rv = P.STFTMaskedNoiseReverb(ir_len=60001, flashfftconv=False).to(dev)
            p = {k: 0.1 * torch.randn(Rr, 2, 193, device=dev) for k in rv.parameter_size()}
            xr, yr = x[:Rr], y[:Rr]
            ms = timeit(lambda: rv.render_into(xr.view(1, Rr, 2, L), yr.view(1, Rr, 2, L), **p), a.iters)
            print(f"reverb  R={Rr} {ms:8.3f} ms  {16 * Rr * L / 1e9 / ms * 1e3:8.1f} GB/s")
            ms = timeit(lambda: rv._ir_and_gain(p["init_log_magnitude"], p["delta_log_magnitude"], None, True), a.iters)
            print(f"  ir synthesis        {ms:8.3f} ms")
        if a.what in ("cfg2", "configs"):   # BASELINE configs[1]: ParametricEqualizer, 1024 mono rows of 10 s
            for N in (4001, 4000):
                R2, L2 = 1024, 480000
                eq = P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=N).to(dev)
                x2 = torch.randn(R2, 1, L2, device=dev)
                p = {k: torch.randn(R2, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
                ms = timeit(lambda: eq(x2, **p), a.iters)
                print(f"cfg2 PEQ fsm_fir_len={N} R={R2} L={L2}: {ms:8.3f} ms  {8 * R2 * L2 / 1e9 / ms * 1e3:8.1f} GB/s  "
                      f"{R2 * L2 / ms * 1e3:.3e} samples/s")
                del x2, eq
        if a.what in ("cfg3", "configs"):   # BASELINE configs[2]: STFTMaskedNoiseReverb, 512 stereo rows of 5 s
            for N in (60001, 60000):
                R3, L3 = 512, 240000
                rv = P.STFTMaskedNoiseReverb(ir_len=N, flashfftconv=False).to(dev)
                x3 = torch.randn(R3, 2, L3, device=dev)
                p = {k: torch.randn(R3, 2, 193, device=dev) for k in rv.parameter_size()}
                ms = timeit(lambda: rv(x3, **p), a.iters)
                print(f"cfg3 reverb ir_len={N} R={R3} L={L3}: {ms:8.3f} ms  {16 * R3 * L3 / 1e9 / ms * 1e3:8.1f} GB/s  "
                      f"{R3 * L3 / ms * 1e3:.3e} samples/s")
                del x3, rv


if __name__ == "__main__":
    main()
