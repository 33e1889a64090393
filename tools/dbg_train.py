#!/usr/bin/env python3
"""Per-stage timing of one training step of the console graph (forward, then every stage of the stage-wise backward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import bench
from grafx_amd.data import convert_to_tensor
from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
from grafx_amd.render import graph as rg
from grafx_amd.utils import create_empty_parameters

dev = torch.device("cuda")
B, L = int(os.environ.get("B", 32)), 131072
G = bench.console_graph()
procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
torch.manual_seed(0)
params = create_empty_parameters(procs, G, std=0.1)
tparams = nn.ParameterDict({t: nn.ParameterDict({k: nn.Parameter(v.clone().to(dev)) for k, v in d.items()}) for t, d in params.items()})
rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
x = torch.randn(B, 32, 2, L, device=dev)

orig = torch.autograd.grad
log = []
def timed_grad(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(*a, **k)
    torch.cuda.synchronize(); log.append(("autograd.grad", (time.perf_counter() - t0) * 1e3))
    return r
def wrap(name, m):
    f = m.forward
    def g(*a, **k):
        if not torch.is_grad_enabled():
            return f(*a, **k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); log.append((name + ".fwd(grad)", (time.perf_counter() - t0) * 1e3))
        return r
    m.forward = g
for k, m in procs.items(): wrap(k, m)
ia = torch.Tensor.index_add_
def timed_ia(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ia(self, *a, **k)
    torch.cuda.synchronize(); log.append(("index_add_", (time.perf_counter() - t0) * 1e3))
    return r
torch.Tensor.index_add_ = timed_ia
for it in range(3):
    log.clear()
    torch.autograd.grad = timed_grad if it == 2 else orig
    for p in tparams.parameters(): p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = render_grafx(procs, x, tparams, rd)[0]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out.square().mean().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms")
print([f"{n} {ms:.2f}" for n, ms in log], "sum", sum(ms for _, ms in log))
