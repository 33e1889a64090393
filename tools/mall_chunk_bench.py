"""python tools/mall_chunk_bench.py [--pmc]: does the equaliser -> compressor hand-off stay in the 256 MiB Infinity Cache when
the two stages alternate over chunks of a few graphs?  

The console's first two stages on the full batch (256 graphs x 32 strips, stereo, L = 131072): `eq` = the 4001-tap
convolution with the source write-through, `compressor` = the fused dynamics + routing-sum kernel.  Whole-batch order
(eq over all graphs, then the compressor over all graphs) against chunked order (eq, compressor per chunk of g graphs).
Prints HIP-event times; with rocprofv3 --pmc FETCH_SIZE around it the dyn kernel's fetch bytes tell whether its input
came from the cache.  `nt=0|1`: GRAFX_PIPE_HSACO builds of the convolution kernel with / without non-temporal stores."""
import sys

import torch

sys.path.insert(0, ".")
from grafx_amd import ops

B, n, C, L, J, N = 256, 32, 2, 131072, 5, 4001
dev = torch.device("cuda")
torch.manual_seed(0)
x = torch.randn(B, n, C, L, device=dev)
buf = torch.empty(B, 3 * n + J, C, L, device=dev)
src, eq, comp, mo = buf[:, :n], buf[:, n:2 * n], buf[:, 2 * n:3 * n], buf[:, 3 * n:]
h = torch.randn(n, 1, N, device=dev) / N ** 0.5
Hs = ops.fir_spectrum(h.reshape(n, N))
p = [torch.randn(n, 1, device=dev) * 0.1 for _ in range(4)]
dests = [list(range(8 * k, 8 * k + 8)) for k in range(4)] + [list(range(n))]
codes, n_acc, _, _ = ops.mix_schedule(dests, n)
sched = torch.tensor(codes, device=dev)
kw = dict(smoother=1, iir_len=16383, knee="quadratic", gate=False, param_rows=n)


def stage_pair(b0, b1):
    ops.fftconv(x[b0:b1], Hs, N, 1, out=eq[b0:b1], tee=src[b0:b1], h_rows=n)
    mix = {"sched": sched, "n_acc": n_acc, "out": mo[b0:b1]}
    ops.dynamics_fused(eq[b0:b1], *p, **kw, out=comp[b0:b1], mix=mix)
    assert mix.get("done")


def whole():
    ops.fftconv(x, Hs, N, 1, out=eq, tee=src, h_rows=n)
    mix = {"sched": sched, "n_acc": n_acc, "out": mo}
    ops.dynamics_fused(eq, *p, **kw, out=comp, mix=mix)
    assert mix.get("done")


def chunked(g):
    def run():
        for b0 in range(0, B, g):
            stage_pair(b0, min(B, b0 + g))
    return run


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


whole()
ref = buf.clone()
cases = [("whole batch", whole)] + [(f"chunks of {g:3d} graphs", chunked(g)) for g in (64, 32, 16, 8, 6, 4, 2, 1)]
if "--pmc" in sys.argv:       # one pass each, for the counters
    for name, fn in cases:
        fn()
    torch.cuda.synchronize()
    sys.exit(0)
for name, fn in cases + cases[:1]:
    ms = timeit(fn)
    same = torch.equal(buf, ref)
    print(f"{name:24s} {ms:8.3f} ms   bit-identical buffer: {same}", flush=True)
