import sys, time, torch
sys.path.insert(0, '.')
import bench
from grafx_amd.data import convert_to_tensor
from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
from grafx_amd.utils import create_empty_parameters
dev = torch.device('cuda')
G = bench.console_graph()
procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
torch.manual_seed(0)
params = {t: {k: v.detach().to(dev) for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
x = torch.randn(256, 32, 2, 131072, device=dev)
rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method='beam')).to(dev)
y = None
for i in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        y = render_grafx(procs, x, params, rd, parameters_grad=False)[0]
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = torch.cuda.memory_stats()
    print(f"step {i}: launch {1e3*(t1-t0):7.1f} ms  total {1e3*(t2-t0):7.1f} ms  reserved {torch.cuda.memory_reserved()/2**30:6.1f} GiB  allocs {st['num_device_alloc']} frees {st['num_device_free']}")
