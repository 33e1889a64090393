#!/usr/bin/env python3
"""Short-FIR matrix-core kernel vs the FFT tile kernel by tap count (run on the GPU box):
    python tools/fir_crossover.py [--rows 2048] [--length 131072]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timeit(fn, iters=5):
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
    ap.add_argument("--rows", type=int, default=2048)
    ap.add_argument("--length", type=int, default=131072)
    a = ap.parse_args()
    from grafx_amd import ops

    R, L = a.rows, a.length
    x = torch.randn(R, 2, L, device="cuda")
    y = torch.empty_like(x)
    gb = 16 * R * L / 1e9
    print(f"# {R} stereo rows x {L} samples, per-row filters; algorithmic bytes {gb:.2f} GB per call")
    print("# taps   fir_mfma ms  (GB/s)    fftconv1 ms  (GB/s)   TFLOP/s(useful 2N)")
    for N in (8, 16, 32, 48, 64, 96, 128, 192, 256, 384, 512):
        h = torch.randn(R, 1, N, device="cuda") / N ** 0.5
        Hs = ops.fir_spectrum(h.reshape(-1, N))
        t_m = timeit(lambda: ops.fir_direct(x, h, out=y))
        t_f = timeit(lambda: ops.fftconv(x, Hs, N, 1, out=y))
        print(f"{N:6d}   {t_m:9.3f}  {gb / t_m * 1e3:8.1f}   {t_f:9.3f}  {gb / t_f * 1e3:8.1f}   {2 * N * 2 * R * L / t_m / 1e9:8.1f}")


if __name__ == "__main__":
    main()
