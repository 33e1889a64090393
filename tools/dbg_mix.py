import sys, torch
sys.path.insert(0, ".")
import bench
from grafx_amd import ops
from grafx_amd.data import convert_to_tensor
from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
from grafx_amd.utils import create_empty_parameters
dev = torch.device("cuda")
G = bench.console_graph(n_ch=8, n_bus=2)
rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
for i in range(rd.max_order + 1):
    st = rd.iter_list[i]
    print(i, st.node_type, st.dest_write.idx)
procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
torch.manual_seed(3)
x = torch.randn(2, 8, 2, 32768, device=dev)
params = {t: {k: v.detach().to(dev) for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
bufs = {}
for flag in (True, False, True):
    ops.MIX_FUSION = flag
    with torch.no_grad():
        out, _, buf = render_grafx(procs, x, params, rd, parameters_grad=False)
    torch.cuda.synchronize()
    if flag in bufs:
        print("repeat equal:", torch.equal(bufs[flag], buf))
    bufs[flag] = buf.clone()
d = (bufs[True] - bufs[False]).abs().amax(dim=(0, 2, 3))
print("per-node max diff:", d.tolist())
