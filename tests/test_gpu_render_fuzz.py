"""Randomised render parity: random DAGs of the headline processor types (plus utility mix nodes), random lengths
(odd and even, shorter and longer than the filters, not multiples of any tile), 3-D and 4-D inputs, both schedulers —
HIP render vs the CPU oracle render of the same graph, and training gradients through the one-node backward."""
import random

import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def random_graph(rng, n_src, n_proc):
    from grafx_amd.data import GRAFX, NodeConfigs

    G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
    nodes = [G.add("in") for _ in range(n_src)]
    for _ in range(n_proc):
        kind = rng.choice(["eq", "compressor", "eq", "compressor", "reverb", "mix"])
        v = G.add(kind)
        for s in rng.sample(nodes, 1 if kind != "mix" else min(len(nodes), rng.randint(2, 4))):
            G.connect(s, v)
        nodes.append(v)
    out = G.add("out")
    for s in rng.sample(nodes[n_src:], min(3, len(nodes) - n_src)):
        G.connect(s, out)
    return G


def build(L, fir, iir, ir):
    import grafx_amd.processors as P
    import oracle

    hip = {"eq": P.ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=fir).cuda(),
           "compressor": P.Compressor(energy_smoother="iir", iir_len=iir, flashfftconv=False).cuda(),
           "reverb": P.STFTMaskedNoiseReverb(ir_len=ir, flashfftconv=False).cuda()}
    ref = {"eq": oracle.OracleParametricEqualizer(num_filters=4, fsm_fir_len=fir),
           "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=iir),
           "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=ir)}
    return hip, ref


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_SEEDS", 12))))
def test_random_graphs_match_the_oracle_render(seed):
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    rng = random.Random(seed)
    torch.manual_seed(seed)
    L = rng.choice([777, 1500, 2048, 4099, 6001])
    fir, iir, ir = rng.choice([(257, 255, 1501), (256, 256, 1500), (513, 1023, 3001)])
    hip, ref = build(L, fir, iir, ir)
    G = random_graph(rng, n_src=rng.randint(1, 3), n_proc=rng.randint(3, 7))
    method = rng.choice(["beam", "greedy", "one-by-one"])
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method=method))
    params = create_empty_parameters(hip, G, std=0.3)
    n_in = len([1 for _, d in G.nodes(data=True) if d["node_type"] == "in"])
    batched = method != "one-by-one" and rng.random() < 0.7   # one-by-one keeps a list buffer: 3-D inputs only
    x = torch.randn(*((rng.randint(1, 3),) if batched else ()), n_in, 2, L)
    with torch.no_grad():
        y_ref, _, buf_ref = render_grafx(ref, x, params, rd)
        p_gpu = {t: {k: v.cuda() for k, v in d.items()} for t, d in params.items()}
        y, _, buf = render_grafx(hip, x.cuda(), p_gpu, rd.to("cuda"))
    if isinstance(buf_ref, list):
        buf_ref, buf = torch.cat([b.cpu() for b in buf_ref], 0), torch.cat([b.cpu() for b in buf], 0)
    assert_close(buf.cpu(), buf_ref, 5e-5, f"seed {seed}: L={L} fir={fir} iir={iir} ir={ir} {method} batched={batched}")
    assert_close(y.cpu(), y_ref, 5e-5, "output")


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_GRAD_SEEDS", 8))))
def test_random_graph_gradients_match_the_oracle(seed):
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    rng = random.Random(100 + seed)
    torch.manual_seed(seed)
    L = rng.choice([1024, 2047])
    hip, ref = build(L, 257, 255, 1501)
    G = random_graph(rng, n_src=2, n_proc=rng.randint(3, 5))
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = create_empty_parameters(hip, G, std=0.3)
    x = torch.randn(2, 2, 2, L)

    def run(procs, dev):
        p = {t: {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in d.items()} for t, d in params.items()}
        xin = x.detach().clone().to(dev).requires_grad_(True)
        y, _, buf = render_grafx(procs, xin, p, rd.to(dev) if dev != "cpu" else rd, input_signal_grad=True)
        w = torch.linspace(0.5, 1.5, L, device=dev)
        loss = (y * w).square().mean()
        if seed % 2 == 1:  # the loss also looks at an intermediate node of the returned signal buffer
            loss = loss + 0.3 * (buf[:, buf.shape[1] // 2] * w).abs().mean()
        loss.backward()
        return xin.grad.cpu(), {(t, k): v.grad.cpu() for t, d in p.items() for k, v in d.items() if v.grad is not None}

    gx_ref, gp_ref = run(ref, "cpu")
    gx, gp = run(hip, "cuda")
    assert_close(gx, gx_ref, 2e-3, "input gradient")
    assert gp_ref and set(gp) == set(gp_ref)
    for key in gp_ref:
        scale = gp_ref[key].abs().max().clamp_min(1e-8)
        assert (gp[key] - gp_ref[key]).abs().max() <= 5e-3 * scale, (key, (gp[key] - gp_ref[key]).abs().max().item(), scale.item())
