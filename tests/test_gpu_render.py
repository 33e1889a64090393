"""GPU parity of full graph renders (render_grafx + HIP processors) vs the reference's outputs."""
import pytest
import torch

from conftest import assert_close
from test_routing_golden import build_cfg1, build_console

pytestmark = pytest.mark.gpu


def test_cfg1_plumbing_graph(golden):
    """BASELINE configs[0]: in -> StereoGain -> BiquadFilter -> out."""
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.processors import BiquadFilter, StereoGain
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render

    g = golden("g8_render")
    procs = {"gain": StereoGain().cuda(), "biquad": BiquadFilter(num_filters=1, flashfftconv=False, fsm_fir_len=257).cuda()}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(build_cfg1()), method="beam")).to("cuda")
    params = {"gain": {"log_gain": g["cfg1_p_gain_log_gain"].cuda()},
              "biquad": {k: g[f"cfg1_p_biquad_{k}"].cuda() for k in ("Bs", "A1_pre", "A2_pre")}}
    with torch.no_grad():
        y, _, buf = render_grafx(procs, g["cfg1_x"].cuda(), params, rd)
    assert_close(y.cpu(), g["cfg1_y"], 1e-5, "cfg1 y")
    assert_close(buf.cpu(), g["cfg1_buf"], 1e-5, "cfg1 buffer")


def test_console8_batched_graph(golden):
    """8-channel / 2-bus console (EQ + compressor + reverb + bus sums), batch 2, vs the reference render."""
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render

    g = golden("g8_render")
    procs = {
        "eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=257).cuda(),
        "compressor": Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False).cuda(),
        "reverb": STFTMaskedNoiseReverb(ir_len=1501, flashfftconv=False).cuda(),
    }
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(build_console(8, 2)), method="beam")).to("cuda")
    params = {t: {k: g[f"console8_p_{t}_{k}"].cuda() for k in procs[t].parameter_size()} for t in procs}
    with torch.no_grad():
        y, _, buf = render_grafx(procs, g["console8_x"].cuda(), params, rd)
    assert_close(y.cpu(), g["console8_y"], 2e-5, "console8 y")
    assert_close(buf[:, -8:].cpu(), g["console8_buf_last8"], 2e-5, "console8 buffer tail")


@pytest.mark.gpu
def test_batch_shared_parameters_equal_expanded_parameters():
    """4-D input: processors that accept un-expanded (per-node) parameters must produce exactly what they produce
    from the B-times expanded copies upstream feeds them (render/graph.py:68-75)."""
    import torch

    import grafx_amd.processors as P

    torch.manual_seed(5)
    B, n, L = 3, 4, 20000
    x4 = torch.randn(B, n, 2, L, device="cuda")
    cases = [
        (P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=1025),
         {k: 0.3 * torch.randn(n, 1, 6, device="cuda") for k in ("w0", "q_inv", "log_gain")}),
        (P.Compressor(energy_smoother="iir", iir_len=1023, flashfftconv=False),
         {"log_threshold": torch.randn(n, 1, device="cuda") - 2, "log_ratio": torch.randn(n, 1, device="cuda"),
          "log_knee": torch.randn(n, 1, device="cuda"), "z_alpha_pre": torch.randn(n, 1, device="cuda") + 2}),
        (P.Compressor(energy_smoother="ballistics", flashfftconv=False),   # a path without native row sharing
         {"log_threshold": torch.randn(n, 1, device="cuda") - 2, "log_ratio": torch.randn(n, 1, device="cuda"),
          "log_knee": torch.randn(n, 1, device="cuda"), "z_alpha_pre": torch.randn(n, 2, device="cuda")}),
        (P.STFTMaskedNoiseReverb(ir_len=3001, flashfftconv=False),
         {"init_log_magnitude": torch.randn(n, 2, 193, device="cuda"), "delta_log_magnitude": torch.randn(n, 2, 193, device="cuda")}),
    ]
    for m, p in cases:
        m = m.cuda()
        assert m.accepts_shared_params
        expanded = {k: v.unsqueeze(0).expand(B, *v.shape).reshape(B * n, *v.shape[1:]).contiguous() for k, v in p.items()}
        a, b = torch.empty_like(x4), torch.empty_like(x4)
        with torch.no_grad():
            m.render_into(x4, a, _shared_rows=n, **p)
            m.render_into(x4, b, **expanded)
        assert torch.equal(a, b), type(m).__name__


@pytest.mark.gpu
def test_prepared_stage_state_gives_the_same_output_as_the_inline_design():
    """prepare() + render_into(_prepared=...) (what the render's side stream does) vs. render_into alone."""
    import torch

    import grafx_amd.processors as P

    torch.manual_seed(9)
    B, n, L = 2, 3, 30000
    x4 = torch.randn(B, n, 2, L, device="cuda")
    cases = [
        (P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=2049),
         {k: 0.3 * torch.randn(n, 1, 6, device="cuda") for k in ("w0", "q_inv", "log_gain")}),
        (P.ParametricEqualizer(num_filters=4, processor_channel="stereo"),   # upstream default lengths / flashfftconv
         {k: 0.3 * torch.randn(n, 2, 4, device="cuda") for k in ("w0", "q_inv", "log_gain")}),
        (P.STFTMaskedNoiseReverb(ir_len=9001, flashfftconv=False),
         {"init_log_magnitude": torch.randn(n, 2, 193, device="cuda"), "delta_log_magnitude": torch.randn(n, 2, 193, device="cuda")}),
    ]
    for m, p in cases:
        m = m.cuda()
        a, b = torch.empty_like(x4), torch.empty_like(x4)
        with torch.no_grad():
            state = m.prepare(_shared_rows=n, **p)
            assert state is not None
            m.render_into(x4, a, _shared_rows=n, _prepared=state, **p)
            m.render_into(x4, b, _shared_rows=n, **p)
        assert torch.equal(a, b), type(m).__name__
    # configurations without a parameter-only split say so
    assert P.ParametricEqualizer(num_filters=3, processor_channel="midside", flashfftconv=False, fsm_fir_len=257).cuda().prepare(
        **{k: torch.zeros(n, 2, 3, device="cuda") for k in ("w0", "q_inv", "log_gain")}) is None


@pytest.mark.gpu
def test_hip_processors_next_to_a_user_defined_torch_processor():
    """A graph that mixes HIP processors with a plain torch module (no render_into): the render takes the generic,
    upstream-shaped loop and the HIP processors are called through their ordinary forward()."""
    import torch

    import grafx_amd.processors as P
    import oracle
    from grafx_amd.data import GRAFX, NodeConfigs, convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    class Tilt(torch.nn.Module):  # a user's own processor, torch ops only
        def forward(self, input_signals, slope):
            ramp = torch.linspace(-1, 1, input_signals.shape[-1], device=input_signals.device)
            return input_signals * (1 + 0.1 * torch.tanh(slope)[..., None] * ramp)

        def parameter_size(self):
            return {"slope": 1}

    G = GRAFX(config=NodeConfigs(["eq", "tilt", "compressor"]))
    a, b = G.add("in"), G.add("in")
    e1, e2, t1, c1, m, out = G.add("eq"), G.add("eq"), G.add("tilt"), G.add("compressor"), G.add("mix"), G.add("out")
    for s, d in ((a, e1), (b, e2), (e1, t1), (e2, c1), (t1, m), (c1, m), (m, out), (t1, out)):
        G.connect(s, d)
    hip = {"eq": P.ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=257).cuda(), "tilt": Tilt(),
           "compressor": P.Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False).cuda()}
    ref = {"eq": oracle.OracleParametricEqualizer(num_filters=4, fsm_fir_len=257), "tilt": Tilt(),
           "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=255)}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    torch.manual_seed(4)
    params = {t: {k: v.detach() for k, v in d.items()} for t, d in create_empty_parameters(hip, G, std=0.3).items()}
    x = torch.randn(3, 2, 2, 3000)
    with torch.no_grad():
        want, _, wbuf = render_grafx(ref, x, params, rd)
        got, _, gbuf = render_grafx(hip, x.cuda(), {t: {k: v.cuda() for k, v in d.items()} for t, d in params.items()},
                                    rd.to("cuda"))
    assert_close(got.cpu(), want, 2e-5, "output")
    assert_close(gbuf.cpu(), wbuf, 2e-5, "signal buffer")

    # and a training step through the same mixed graph (the taped generic loop around the HIP autograd functions)
    def grads(procs, dev):
        p = {t: {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in d.items()} for t, d in params.items()}
        rd_ = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)  # .to() moves in place
        y, _, _ = render_grafx(procs, x.to(dev), p, rd_)
        y.square().mean().backward()
        return {(t, k): v.grad.cpu() for t, d in p.items() for k, v in d.items()}

    g_ref, g_hip = grads(ref, "cpu"), grads(hip, "cuda")
    assert set(g_ref) == set(g_hip)
    for key, wv in g_ref.items():
        assert (g_hip[key] - wv).abs().max() <= 5e-3 * wv.abs().max().clamp_min(1e-8), key


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_output_only_render_gives_the_same_output_bits(seed):
    """render_grafx(keep_signal_buffer=False) (an extension: upstream always returns the buffer) must not change one bit
    of the output node -- on the console (where the channel strips' rows are then never stored and the sources never
    copied) and on random graphs (where some stage usually DOES read the rows again, or sums take source rows)."""
    import random

    import bench
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters
    from test_gpu_render_fuzz import build, random_graph

    rng = random.Random(seed)
    torch.manual_seed(seed)
    if seed == 0:
        G = bench.console_graph()
        procs = {k: v.cuda() for k, v in bench.hip_processors(lens=dict(fsm_fir_len=513, iir_len=1023, ir_len=3001)).items()}
        x = torch.randn(2, 32, 2, 8192, device="cuda")
    else:
        G = random_graph(rng, n_src=rng.randint(1, 3), n_proc=rng.randint(3, 8))
        procs, _ = build(0, 257, 255, 1501)
        n_in = len([1 for _, d in G.nodes(data=True) if d["node_type"] == "in"])
        x = torch.randn(rng.randint(1, 3), n_in, 2, rng.choice([2048, 4096, 6000]), device="cuda")
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")
    params = {t: {k: v.detach().cuda() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.3).items()}
    with torch.no_grad():
        y_full, _, buf = render_grafx(procs, x, params, rd, parameters_grad=False)
        y_lean, _, none = render_grafx(procs, x, params, rd, parameters_grad=False, keep_signal_buffer=False)
    assert none is None and buf is not None
    assert torch.equal(y_full, y_lean)


def test_a_cuda_render_off_the_buffer_path_says_so(golden):
    """A render of CUDA signals that cannot take the in-place buffer path (here: the 'one-by-one' schedule) runs upstream's
    loop around the HIP processors -- and says why (GenericRenderPathWarning) instead of silently being slow; the result is
    the fast path's."""
    import warnings

    from grafx_amd.data import convert_to_tensor
    from grafx_amd.processors import BiquadFilter, StereoGain
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.render.graph import GenericRenderPathWarning

    g = golden("g8_render")
    procs = {"gain": StereoGain().cuda(), "biquad": BiquadFilter(num_filters=1, flashfftconv=False, fsm_fir_len=257).cuda()}
    params = {"gain": {"log_gain": g["cfg1_p_gain_log_gain"].cuda()},
              "biquad": {k: g[f"cfg1_p_biquad_{k}"].cuda() for k in ("Bs", "A1_pre", "A2_pre")}}
    x = g["cfg1_x"].cuda()
    fast = prepare_render(reorder_for_fast_render(convert_to_tensor(build_cfg1()), method="beam")).to("cuda")
    slow = prepare_render(reorder_for_fast_render(convert_to_tensor(build_cfg1()), method="one-by-one")).to("cuda")
    with torch.no_grad():
        with warnings.catch_warnings():
            warnings.simplefilter("error", GenericRenderPathWarning)
            y_fast = render_grafx(procs, x, params, fast)[0]
        with pytest.warns(GenericRenderPathWarning, match="one-by-one"):
            y_slow = render_grafx(procs, x[0] if slow.method == "one-by-one" and x.ndim == 4 else x, params, slow)[0]
    assert_close(y_slow.reshape(y_fast.shape).cpu(), y_fast.cpu(), 1e-6, "one-by-one vs beam")
