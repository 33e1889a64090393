"""python tools/prepare_ab.py: the headline step with the later stages' parameter-only work (filter design, the reverb's
impulse response) (a) on a side stream under the first equaliser stage, (b) under the compressor stage, (c) inline on the
main stream -- whole-step time and the per-kernel HIP-event times of the signal kernels."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench
from grafx_amd import ops
from grafx_amd.render import graph as rg

dev = torch.device("cuda")
step = bench.console_case(torch, dev, 256, 131072, bench.LENS)
print("| mode | ms/step (40 steps) | " + " | ".join(["pipe_t1_o8", "pipe_t0_o8", "dyn kernels", "xspec + macinv"]) + " |")
print("|---|---|---|---|---|---|")
for rep in range(2):
    for mode in ("under_first", "under_second", "inline"):
        rg.PREPARE_MODE = mode
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        with ops.profiling() as prof:
            t0 = time.perf_counter()
            for _ in range(40):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 40 * 1e3
        ms = {k: sum(a.elapsed_time(b) for a, b, _ in v) / 40 for k, v in prof.items()}
        # records are keyed by the kernels' own names (gfx_fftconv_last_kernel / gfx_dynamics_last_kernel): sum by prefix
        tot = lambda pre: sum(v for k, v in ms.items() if k.startswith(pre))  # noqa: E731
        cols = [ms.get("gfx_fftconv_pipe_t1_o8", 0), ms.get("gfx_fftconv_pipe_t0_o8", 0), tot("dyn_"), tot("xspec_kernel+")]
        print(f"| {mode} | {dt:.3f} | " + " | ".join(f"{c:.3f}" for c in cols) + " |", flush=True)
