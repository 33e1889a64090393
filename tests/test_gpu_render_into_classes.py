"""Every processor class that defines render_into (the in-place buffer path of render_grafx) is rendered through
render_grafx itself -- 3-D and 4-D input, with and without gradients -- and compared with the same processor called
through its ordinary forward() on gathered rows (the generic, upstream-shaped loop semantics).  Round-1 advisor
finding: ApproxCompressor inherited render_into but its forward() did not take the `_shared_rows` argument it passes,
so it could never run inside render_grafx on the GPU; direct forward() tests do not see that class of bug."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cases():
    import grafx_amd.processors as P

    mk = lambda *shape, s=0.3, o=0.0: s * torch.randn(*shape) + o  # noqa: E731
    return [
        ("peq", lambda: P.ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=513),
         lambda n: {k: mk(n, 1, 4) for k in ("w0", "q_inv", "log_gain")}),
        ("compressor", lambda: P.Compressor(energy_smoother="iir", iir_len=511, flashfftconv=False),
         lambda n: {"log_threshold": mk(n, 1, o=-2), "log_ratio": mk(n, 1), "log_knee": mk(n, 1), "z_alpha_pre": mk(n, 1, o=2)}),
        ("noisegate", lambda: P.NoiseGate(energy_smoother="iir", iir_len=511, flashfftconv=False),
         lambda n: {"log_threshold": mk(n, 1, o=-2), "log_ratio": mk(n, 1), "log_knee": mk(n, 1), "z_alpha_pre": mk(n, 1, o=2)}),
        ("approx_compressor", lambda: P.ApproxCompressor(iir_len=511, flashfftconv=False),
         lambda n: {"z_alpha": mk(n, 1, o=2), "log_threshold": mk(n, 1, o=-2), "log_ratio": mk(n, 1), "log_knee": mk(n, 1)}),
        ("reverb", lambda: P.STFTMaskedNoiseReverb(ir_len=1501, flashfftconv=False),
         lambda n: {"init_log_magnitude": mk(n, 2, 193, s=1.0), "delta_log_magnitude": mk(n, 2, 193, s=1.0)}),
        ("biquad", lambda: P.BiquadFilter(num_filters=2, flashfftconv=False, fsm_fir_len=257),
         lambda n: {"Bs": mk(n, 2, 3), "A1_pre": mk(n, 2), "A2_pre": mk(n, 2)}),
        ("gain", lambda: P.StereoGain(), lambda n: {"log_gain": mk(n, 2)}),
        ("tanh", lambda: P.TanhDistortion(), None),
    ]


@pytest.mark.parametrize("name", [c[0] for c in _cases()])
@pytest.mark.parametrize("ndim", [3, 4])
@pytest.mark.parametrize("grad", [False, True])
def test_render_into_class_through_render_grafx(name, ndim, grad):
    from grafx_amd.data import GRAFX, NodeConfigs, convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    _, make, make_params = next(c for c in _cases() if c[0] == name)
    torch.manual_seed(11)
    proc = make().cuda()
    assert hasattr(proc, "render_into")
    G = GRAFX(config=NodeConfigs(["fx"]))
    out_id = G.add("out")
    n = 3
    for _ in range(n):
        _, last = G.add_serial_chain(["in", "fx"])
        G.connect(last, out_id)
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")
    procs = {"fx": proc}
    if make_params is None:
        p = {k: v.detach() for k, v in create_empty_parameters(procs, G, std=0.3)["fx"].items()}
    else:
        p = make_params(n)
    L = 3000
    x = torch.randn(*((2,) if ndim == 4 else ()), n, 2, L, device="cuda")
    params = {"fx": {k: v.cuda().requires_grad_(grad) for k, v in p.items()}}
    y, _, buf = render_grafx(procs, x, params, rd)   # in-place buffer path (render_into)
    # the same nodes through forward() on plain rows, parameters expanded over the batch as upstream does
    ref_params = {k: v.detach().clone().requires_grad_(grad) for k, v in params["fx"].items()}
    B = 2 if ndim == 4 else 1
    rows = x.reshape(B * n, 2, L)
    exp = {k: v.unsqueeze(0).expand(B, *v.shape).reshape(B * n, *v.shape[1:]) for k, v in ref_params.items()}
    fx = proc(rows, **exp)
    fx = fx[0] if isinstance(fx, tuple) else fx
    want = fx.view(B, n, 2, L).sum(1, keepdim=True)
    got = y.view(B, 1, 2, L)
    tol = 1e-6   # (the out node's sum runs in the gather kernel here and in torch there: same values, other order)
    assert (got - want).abs().max() <= tol * want.abs().max(), name
    if grad:
        got.square().mean().backward()
        want.square().mean().backward()
        for k in ref_params:
            a, b = params["fx"][k].grad, ref_params[k].grad
            assert a is not None and b is not None, k
            assert (a - b).abs().max() <= 5e-4 * b.abs().max().clamp_min(1e-12), (name, k)
