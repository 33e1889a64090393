"""python tools/sustained_probe.py: the headline step in chunks of 50 (synchronised per chunk) with and without the hwmon
sampler thread -- does reading the clock / power files 20 times a second slow the render down?"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench

dev = torch.device("cuda")
step = bench.console_case(torch, dev, 256, 131072, bench.LENS)
for _ in range(5):
    step()
torch.cuda.synchronize()
for mode in ("no sampler", "sampler 50 ms", "no sampler", "sampler 250 ms"):
    smp = None
    if mode.startswith("sampler"):
        smp = bench.GpuSampler(period=0.05 if "50" in mode else 0.25)
        smp.__enter__()
    chunks = []
    for c in range(6):
        t0 = time.perf_counter()
        for _ in range(50):
            y = step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        chunks.append(((time.perf_counter() - t0) / 50 * 1e3, t_host / 50 * 1e3))
    if smp is not None:
        smp.__exit__(None, None, None)
    print(mode, "ms/step per chunk of 50 (host enqueue ms/step):", " ".join(f"{a:.2f}({b:.2f})" for a, b in chunks),
          "" if smp is None else {k: (round(v, 1) if isinstance(v, float) else v) for k, v in smp.summary().items() if k in ("sclk_mhz_mean", "power_w_mean", "samples")}, flush=True)
