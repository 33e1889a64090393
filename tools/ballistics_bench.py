"""Time of the ballistics recursion (csrc/ballistics.hip) at the console's sizes:
   python tools/ballistics_bench.py [--rows 9216 1024 256] [--length 131072] [--iters 5] [--old-lib grafx_amd/lib/r4base.so]
-> ms per call and GB/s over the algorithmic bytes (read every input sample once, write every output sample once), for
   * z ~ randn * 0.1 (bench.py's parameter scale: coefficients ~ 0.5, rows cut into verified chunks),
   * z = -6 (coefficients 2.5e-3: a warm-up longer than a chunk, every row walked whole by the second launch),
   both schedules, the energy-source form on stereo rows, and -- with --old-lib -- round 4's one-wave-per-64-rows kernel
   through the same entry point of that library."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from grafx_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, nargs="+", default=[9216, 1024, 256])
ap.add_argument("--length", type=int, default=131072)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--old-lib", default=None)
a = ap.parse_args()
old = ctypes.CDLL(a.old_lib) if a.old_lib and os.path.exists(a.old_lib) else None


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


print("| rows | z_alpha | form | ms | GB/s (algorithmic) | of 8 TB/s |")
print("|---|---|---|---|---|---|")
L = a.length
for R in a.rows:
    torch.manual_seed(0)
    u = torch.rand(R, L, device="cuda") * 2
    x = torch.randn(R, 2, L, device="cuda")
    y = torch.empty_like(u)
    for zname, z in (("randn*0.1", torch.randn(R, 2, device="cuda") * 0.1), ("-6", torch.full((R, 2), -6.0, device="cuda"))):
        rows = []
        for sched in ("chunks", "rows"):
            rows.append((f"(R, L) rows, {sched}", timed(lambda: ops.ballistics(u, z, schedule=sched)), 8 * R * L))
        rows.append(("energy of (R, 2, L), chunks", timed(lambda: ops.ballistics_energy(x, z)), 12 * R * L))
        if old is not None:
            vp = ctypes.c_void_p
            s = torch.cuda.current_stream().cuda_stream
            rows.append(("round-4 kernel", timed(lambda: old.gfx_ballistics_f32(vp(u.data_ptr()), vp(z.data_ptr()), vp(y.data_ptr()),
                                                                              ctypes.c_int64(R), ctypes.c_int64(L), vp(s))),
                         8 * R * L))
        for name, ms, nbytes in rows:
            gbs = nbytes / ms / 1e6
            print(f"| {R} | {zname} | {name} | {ms:.3f} | {gbs:.0f} | {gbs / 8000:.3f} |", flush=True)

# the adjoint (training): rows cut into chunks with a 2048-sample warm-up against one lane per row
print()
print("| rows | z_alpha | adjoint form | ms | GB/s (algorithmic: x, y, g in, gx out) | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for R in a.rows:
    torch.manual_seed(1)
    u = torch.rand(R, L, device="cuda") * 2
    g = torch.randn(R, L, device="cuda")
    for zname, z in (("randn*0.1", torch.randn(R, 2, device="cuda") * 0.1), ("-6", torch.full((R, 2), -6.0, device="cuda"))):
        y = ops.ballistics(u, z)
        for sched in ("chunks", "rows"):
            ms = timed(lambda: ops.ballistics_bwd(u, y, g, z, schedule=sched))
            gbs = 16 * R * L / ms / 1e6
            print(f"| {R} | {zname} | {sched} | {ms:.3f} | {gbs:.0f} | {gbs / 8000:.3f} |", flush=True)
