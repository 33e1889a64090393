"""Experiment: the whole training step (forward + stage-wise backward) captured into one HIP graph.

    python tools/experiments/train_graph/train_graph.py [graphs_per_gpu]
"""
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from grafx_amd.data import convert_to_tensor  # noqa: E402
from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render  # noqa: E402
from grafx_amd.utils import create_empty_parameters  # noqa: E402

Bt = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = 131072
dev = torch.device("cuda", 0)
G = bench.console_graph()
rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
torch.manual_seed(1234)
params = {t: {k: v.detach().to(dev) for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
tparams = nn.ParameterDict({t: nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in d.items()}) for t, d in params.items()})
plist = list(tparams.parameters())
x = torch.randn(Bt, 32, 2, L, device=dev)


def step():
    for p in plist:
        p.grad = None
    out = render_grafx(procs, x, tparams, rd)[0]
    out.square().mean().backward()


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"eager   {timed(step):8.2f} ms/step at {Bt} graphs")
eager_grads = [p.grad.clone() for p in plist]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
for p in plist:
    p.grad = None
with torch.cuda.graph(g, capture_error_mode=os.environ.get("CAPTURE_MODE", "thread_local")):
    out = render_grafx(procs, x, tparams, rd)[0]
    out.square().mean().backward()
print(f"replay  {timed(g.replay):8.2f} ms/step")
worst = max(((p.grad - e).abs().max() / e.abs().max().clamp_min(1e-12)).item() for p, e in zip(plist, eager_grads))
print(f"gradients vs eager: worst relative difference {worst:.2e}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
