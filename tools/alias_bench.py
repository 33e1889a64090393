"""Time of the odd-length aliasing (ops.odd_alias) one row per transform (csrc/czt.hip) against two rows per transform
(csrc/czt_pair.hip):  python tools/alias_bench.py [--rows 4096] [--P 135071 147455 191071] [--iters 5]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grafx_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4096)
ap.add_argument("--P", type=int, nargs="+", default=[135071, 147455, 191071])
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


print("| P | precision | rows | one row / transform ms | two rows / transform ms | ratio |")
print("|---|---|---|---|---|---|")
for P in a.P:
    z = torch.randn(a.rows, P, device="cuda")
    for precise in (False, True):
        rows = a.rows // (2 if precise else 1)
        t = []
        for pairs in (False, True):
            ops.ALIAS_PAIRS = pairs
            t.append(timed(lambda: ops.odd_alias(z[:rows], 0, 131072, precise=precise)))
        print(f"| {P} | {'double' if precise else 'float'} | {rows} | {t[0]:.2f} | {t[1]:.2f} | {t[1] / t[0]:.3f} |", flush=True)
