import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import bench
from grafx_amd.processors import Compressor
from grafx_amd.render import graph as render_graph
from grafx_amd.utils import create_empty_parameters
from test_gpu_autograd import _console_gradients
dev = torch.device("cuda")
G = bench.console_graph(n_ch=8, n_bus=2)
procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
procs["compressor"] = Compressor(energy_smoother="iir", iir_len=16383, flashfftconv=False).to(dev)
torch.manual_seed(7)
x = torch.randn(3, 8, 2, 16384, device=dev)
params = create_empty_parameters(procs, G, std=0.1).to(dev)
names = [n for n, _ in params.named_parameters()] + ["gx"]
got = _console_gradients(procs, G, x, params, want_gx=True)
render_graph.BLOCK_FAN_ADJOINT = False
want = _console_gradients(procs, G, x, params, want_gx=True)
want2 = _console_gradients(procs, G, x, params, want_gx=True)
for n, a, b, c in zip(names, got, want, want2):
    print(n, tuple(a.shape), "block vs expanded", float((a - b).abs().max()), "rel", float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), "expanded twice", float((b - c).abs().max()))
