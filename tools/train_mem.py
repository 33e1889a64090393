"""python tools/train_mem.py [batch]: what is alive at the memory peak of one training step of the headline console graph
(forward + backward at `batch` graphs): the caching allocator's history is recorded, replayed to the peak, and the live
blocks are listed with the Python frame that allocated them."""
import sys

import torch

sys.path.insert(0, ".")
import torch.nn as nn

import bench
from grafx_amd.data import convert_to_tensor
from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
from grafx_amd.utils import create_empty_parameters

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 131072
dev = torch.device("cuda")
G = bench.console_graph()
procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
torch.manual_seed(0)
params = create_empty_parameters(procs, G, std=0.1).to(dev)
x = torch.randn(B, 32, 2, L, device=dev)
rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)


def step():
    for p in params.parameters():
        p.grad = None
    render_grafx(procs, x, params, rd)[0].square().mean().backward()


step()
torch.cuda.synchronize()
torch.cuda.empty_cache()
base = torch.cuda.memory_allocated()
torch.cuda.reset_peak_memory_stats()
torch.cuda.memory._record_memory_history(max_entries=200000)
step()
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
print(f"resident before the step (input batch, parameters, constants): {base / 2**30:.2f} GiB; "
      f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
live, cur, peak, peak_live = {}, 0, 0, {}
for ev in snap["device_traces"][0]:
    if ev["action"] == "alloc":
        live[ev["addr"]] = ev
        cur += ev["size"]
        if cur > peak:
            peak, peak_live = cur, dict(live)
    elif ev["action"] in ("free_requested", "free"):
        e = live.pop(ev["addr"], None)
        if e is not None:
            cur -= e["size"]


def where(ev):
    fr = [f for f in ev.get("frames", []) if "/repo/" in f["filename"] and "train_mem" not in f["filename"]]
    return " <- ".join(f"{f['filename'].split('/repo/')[-1]}:{f['line']} {f['name']}" for f in fr[:3]) or "?"


print(f"allocated inside the step at its peak: {peak / 2**30:.2f} GiB in {len(peak_live)} blocks; the largest:")
for ev in sorted(peak_live.values(), key=lambda e: -e["size"])[:14]:
    print(f"  {ev['size'] / 2**30:7.2f} GiB  {where(ev)}")
