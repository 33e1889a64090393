"""Does the chirp-z chain run faster when its workspace stays inside the 256 MiB Infinity Cache?
   python tools/compat_cap_sweep.py [--caps-mib 4096 1024 512 256 192 128 64]
Times bench.py's compat legs (upstream's default even tap counts: every convolve() takes the odd-length aliasing path) with the
transient workspace of one launch chain capped at each size (ops.set_alias_workspace_cap): a smaller cap means more,
shorter chains of the same five kernels over a buffer the next kernel may still find on the die."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from grafx_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--caps-mib", type=int, nargs="+", default=[4096, 1024, 512, 256, 192, 128, 64])
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
lens = bench.REFERENCE_DEFAULT_LENS


def timed(step):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


cases = {"cfg4_compat": bench.console_case(torch, dev, 64, 131072, lens)}
for cfg in ("cfg2", "cfg3"):
    cases[cfg + "_compat"] = bench.processor_case(cfg, torch, dev, 0, lens=lens)[0]
print("| cap MiB | " + " | ".join(cases) + " |")
print("|---|" + "---|" * len(cases))
for cap in a.caps_mib:
    ops.set_alias_workspace_cap(cap << 20)
    print(f"| {cap} | " + " | ".join(f"{timed(s):.2f}" for s in cases.values()) + " |", flush=True)
