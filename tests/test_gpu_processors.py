"""GPU parity of the HIP processors against outputs of the reference itself
(tests/golden/*.npz) — every case goes nn.Module -> ops -> C ABI -> HIP kernels.

Tolerance: 1e-5 on max|y-ref|/max|ref| and on relative L2 (north star: 1e-5 relative fp32;
metric per SURVEY.md H6).  Cases with pre-activation std=1 can place poles close to the unit
circle where the reference's own complex64 response is only ~1e-5 accurate (H6); those use
the stated looser bound and are additionally checked against a float64 evaluation.
"""
import pytest
import torch

import oracle
from conftest import assert_close, assert_parity, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def cu(t):
    return t.cuda() if isinstance(t, torch.Tensor) else t


def run(module, x, params):
    module = module.cuda()
    with torch.no_grad():
        return module(cu(x), **{k: cu(v) for k, v in params.items()}).cpu()


@pytest.mark.parametrize("N", [256, 257, 500])
@pytest.mark.parametrize("K", [1, 6])
def test_iir_fsm(golden, N, K):
    from grafx_amd.processors import IIRFilter

    g = golden("g2_iir_fsm")
    tag = f"N{N}_K{K}"
    flt = IIRFilter(order=2, backend="fsm", flashfftconv=False, fsm_fir_len=N)
    with torch.no_grad():
        fir = flt.fsm_fir(g[f"Bs_{tag}"].cuda(), g[f"As_{tag}"].cuda()).cpu()
    assert_close(fir, g[f"fir_{tag}"], TOL, "fsm fir")
    y = run(flt, g[f"x_{tag}"], {"Bs": g[f"Bs_{tag}"], "As": g[f"As_{tag}"]})
    assert_close(y, g[f"y_{tag}"], TOL, "fsm y")


@pytest.mark.parametrize("ch", ["mono", "stereo", "midside"])
@pytest.mark.parametrize("N", [256, 257])
@pytest.mark.parametrize("std", [0.01, 1.0])
def test_peq(golden, ch, N, std):
    from grafx_amd.processors import ParametricEqualizer

    g = golden("g3_peq")
    tag = f"{ch}_N{N}_std{std}"
    m = ParametricEqualizer(num_filters=6, processor_channel=ch, flashfftconv=False, fsm_fir_len=N)
    p = {k: g[f"{k}_{tag}"] for k in ("w0", "q_inv", "log_gain")}
    y = run(m, g[f"x_{tag}"], p)
    assert_close(y, g[f"y_{tag}"], TOL if std < 1 else 3e-5, "peq")
    if std == 1.0:  # tie-breaker: float64 evaluation of the same formulas
        m64 = oracle.OracleParametricEqualizer(num_filters=6, processor_channel=ch, fsm_fir_len=N)
        y64 = m64(g[f"x_{tag}"].double(), **{k: v.double() for k, v in p.items()})
        ours, ref = rel_err(y, y64)[0], rel_err(g[f"y_{tag}"], y64)[0]
        assert ours <= max(3 * ref, TOL), f"vs float64: ours {ours:.2e}, reference {ref:.2e}"


@pytest.mark.parametrize("K", [1, 4])
@pytest.mark.parametrize("N", [256, 257])
@pytest.mark.parametrize("normalized", [False, True])
def test_biquad(golden, K, N, normalized):
    from grafx_amd.processors import BiquadFilter

    g = golden("g4_biquad_gain")
    tag = f"K{K}_N{N}_norm{int(normalized)}"
    m = BiquadFilter(num_filters=K, normalized=normalized, flashfftconv=False, fsm_fir_len=N)
    p = {k: g[f"{k}_{tag}"] for k in m.parameter_size()}
    # 1e-5 against the reference's own output; where the reference's complex64 response is itself farther than that
    # from a float64 evaluation of the same formulas (poles near the unit circle), the float64 tie-breaker decides
    y64 = oracle.OracleBiquadFilter(num_filters=K, normalized=normalized, fsm_fir_len=N)(
        g[f"x_{tag}"].double(), **{k: v.double() for k, v in p.items()})
    assert_parity(run(m, g[f"x_{tag}"], p), g[f"y_{tag}"], y64.float(), TOL, "biquad")


def test_stereo_gain(golden):
    from grafx_amd.processors import StereoGain

    g = golden("g4_biquad_gain")
    y = run(StereoGain(), g["gain_x"], {"log_gain": g["gain_log_gain"]})
    assert_close(y, g["gain_y"], 1e-6, "gain")


@pytest.mark.parametrize("ir_len", [3000, 3001])
@pytest.mark.parametrize("ch", ["pseudo_midside", "midside", "stereo"])
def test_reverb(golden, ir_len, ch):
    from grafx_amd.processors import STFTMaskedNoiseReverb

    g = golden("g5_reverb")
    tag = f"ir{ir_len}_{ch}"
    m = STFTMaskedNoiseReverb(ir_len=ir_len, processor_channel=ch, flashfftconv=False)
    p = {k: g[f"{k}_{tag}"] for k in ("init_log_magnitude", "delta_log_magnitude")}
    assert_close(run(m, g[f"x_{tag}"], p), g[f"y_{tag}"], TOL, "reverb y")
    if ch == "pseudo_midside":
        with torch.no_grad():
            ir = m.cuda().compute_ir(**{k: v.cuda() for k, v in p.items()}).cpu()
        assert_close(ir, g[f"ir_{tag}"], TOL, "reverb ir")


def test_reverb_gain_envelope(golden):
    from grafx_amd.processors import STFTMaskedNoiseReverb

    g = golden("g5_reverb")
    m = STFTMaskedNoiseReverb(ir_len=3001, gain_envelope=True, flashfftconv=False)
    p = {k: g[f"{k}_genv"] for k in m.parameter_size()}
    assert_close(run(m, g["x_genv"], p), g["y_genv"], TOL, "reverb genv")


@pytest.mark.parametrize("cls", ["Compressor", "NoiseGate"])
@pytest.mark.parametrize("knee", ["hard", "quadratic", "exponential"])
@pytest.mark.parametrize("sm,iir_len", [("iir", 512), ("iir", 511), (None, 0), ("ballistics", 0)])
def test_dynamics(golden, cls, knee, sm, iir_len):
    import grafx_amd.processors as P

    g = golden("g6_dynamics")
    tag = f"{cls}_{knee}_{sm}_{iir_len}"
    kw = dict(energy_smoother=sm, knee=knee, flashfftconv=False)
    if sm == "iir":
        kw["iir_len"] = iir_len
    m = getattr(P, cls)(**kw)
    p = {k: g[f"{k}_{tag}"] for k in m.parameter_size()}
    y = run(m, g["x_shared"], p)
    if sm == "ballistics":  # no float64 tie-breaker needed: same recursion, no FFT noise upstream
        return assert_close(y, g[f"y_{tag}"], TOL, tag)
    M64 = oracle.OracleCompressor if cls == "Compressor" else oracle.OracleNoiseGate
    y64 = M64(energy_smoother=sm, knee=knee, iir_len=iir_len)(g["x_shared"].double(), **{k: v.double() for k, v in p.items()})
    assert_parity(y, g[f"y_{tag}"], y64, TOL, tag)


@pytest.mark.parametrize("gs,in_log", [("iir", False), ("iir", True), ("ballistics", False)])
def test_gain_smoothers(golden, gs, in_log):
    from grafx_amd.processors import Compressor

    g = golden("g6_dynamics")
    tag = f"gs_{gs}_{int(in_log)}"
    m = Compressor(energy_smoother="iir", gain_smoother=gs, gain_smooth_in_log=in_log, iir_len=511, flashfftconv=False)
    p = {k: g[f"{k}_{tag}"] for k in m.parameter_size()}
    assert_close(run(m, g[f"x_{tag}"], p), g[f"y_{tag}"], TOL, tag)


@pytest.mark.parametrize("n", [512, 511])
def test_one_pole(golden, n):
    from grafx_amd.processors import TruncatedOnePoleIIRFilter

    g = golden("g7_g9_smoothers")
    f = TruncatedOnePoleIIRFilter(iir_len=n, flashfftconv=False)
    with torch.no_grad():
        h = f.compute_impulse(g["onepole_z"].cuda()).cpu()
        y = f(g[f"onepole_u_{n}"].cuda(), g["onepole_z"].cuda()).cpu()
    assert_close(h, g[f"onepole_h_{n}"], 1e-6, "one-pole taps")
    assert_close(y, g[f"onepole_y_{n}"], TOL, "one-pole y")


def test_ballistics_provisional(golden):
    from grafx_amd.processors import Ballistics

    g = golden("g7_g9_smoothers")
    with torch.no_grad():
        y = Ballistics()(g["ball_u"].cuda(), g["ball_z"].cuda()).cpu()
    assert_close(y, g["ball_y"], TOL, "ballistics (recalled torchcomp semantics)")


@pytest.mark.parametrize("L,N", [(1024, 128), (1025, 128), (1024, 127), (1000, 301)])
@pytest.mark.parametrize("mode", ["causal", "zerophase"])
def test_convolve_reference_semantics_both_parities(golden, L, N, mode):
    from grafx_amd.processors import convolve

    g = golden("g1_convolve")
    for C, Cf in [(1, 1), (2, 1), (1, 2), (2, 2)]:
        x, h = g[f"x_L{L}_N{N}"][:, :C].contiguous(), g[f"h_L{L}_N{N}"][:, :Cf].contiguous()
        with torch.no_grad():
            y = convolve(x.cuda(), h.cuda(), mode=mode).cpu()
        assert_close(y, g[f"y_{mode}_L{L}_N{N}_C{C}_Cf{Cf}"], TOL, f"convolve C{C} Cf{Cf}")


def test_loud_failures():
    import grafx_amd.processors as P

    m = P.StereoGain()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 8), torch.zeros(1, 2))  # CPU tensors are refused
    y = m(torch.ones(1, 2, 8).cuda(), torch.zeros(1, 2, device="cuda", requires_grad=True))
    assert y.requires_grad  # gradient requested -> differentiable path
    # (the upstream "ssm" quirk for K > 1 used to be forward-only and raised here; it trains since round 6:
    # tests/test_gpu_recursive_iir.py::test_ssm_quirk_with_several_sections_trains)
    ys = P.IIRFilter(backend="ssm", flashfftconv=False)(torch.rand(1, 1, 64).cuda(),
                                                         torch.ones(1, 1, 2, 3, device="cuda", requires_grad=True),
                                                         torch.tensor([1.0, 0.1, 0.1], device="cuda").expand(1, 1, 2, 3))
    assert ys.requires_grad
    with pytest.raises(NotImplementedError):  # the recursive kernel runs first- and second-order sections only
        P.IIRFilter(order=3, backend="lfilter", flashfftconv=False)
    with pytest.raises(ValueError):
        P.IIRFilter(backend="nope", flashfftconv=False)
    with pytest.raises(ValueError):
        P.Compressor(knee="soft")


@pytest.mark.parametrize("N", [8192, 16384])
@pytest.mark.parametrize("K", [1, 6])
def test_iir_fsm_native_taps_for_power_of_two_lengths_above_4096(N, K):
    """fsm_fir_len = 8192 / 16384 (upstream tests/processors/test_filter.py:27): the tile's own inverse real transform
    (gfx_iir_fsm_fir_f32 without a plan) against the oracle's complex64 response + irfft, float64 tie-breaker."""
    from grafx_amd import ops
    from grafx_amd.processors import IIRFilter
    from oracle import lti

    assert ops.iir_fsm_native(N) and not ops.iir_fsm_native(N - 2) and not ops.iir_fsm_native(5000)
    torch.manual_seed(N + K)
    R, Cf = 3, 2
    w0, q, g = (0.6 * torch.randn(R, Cf, K) for _ in range(3))
    Bs, As = oracle.peq_biquad_coefficients(w0, q, g)
    flt = IIRFilter(order=2, backend="fsm", flashfftconv=False, fsm_fir_len=N)
    with torch.no_grad():
        fir = flt.fsm_fir(Bs.cuda(), As.cuda()).cpu()
    ref = lti.iir_fsm_fir(Bs, As, N)
    ref64 = lti.iir_fsm_fir(Bs.double(), As.double(), N)
    assert fir.shape == ref.shape == (R, Cf, N)
    assert_parity(fir, ref, ref64.float(), TOL, f"fsm taps N={N} K={K}")


@pytest.mark.parametrize("order,N", [(1, 512), (3, 513), (4, 1000)])
def test_iir_fsm_other_orders(order, N):
    """IIRFilter(order != 2, backend="fsm") (core/iir.py:96-152 with delays = arange(order + 1); no processor of the package
    uses it): response by torch ops on the GPU, taps by the direct-sum inverse DFT, convolution native -- vs the oracle,
    forward and gradients."""
    from grafx_amd.processors import IIRFilter
    from oracle import lti

    torch.manual_seed(10 * order + N)
    R, Cf, K, L = 3, 1, 2, 3000
    Bs = 0.3 * torch.randn(R, Cf, K, order + 1)
    As = 0.1 * torch.randn(R, Cf, K, order + 1)
    As[..., 0] = 1.0
    x = torch.randn(R, 2, L)
    flt = IIRFilter(order=order, backend="fsm", flashfftconv=False, fsm_fir_len=N)
    h_ref = lti.iir_fsm_fir(Bs, As, N)
    y_ref = lti.convolve(x, h_ref, "causal")
    with torch.no_grad():
        y = flt(x.cuda(), Bs.cuda(), As.cuda()).cpu()
    assert_close(y, y_ref, TOL, f"fsm order {order}")
    Bg, Ag, xg = Bs.cuda().requires_grad_(), As.cuda().requires_grad_(), x.cuda().requires_grad_()
    w = torch.randn(R, 2, L)
    (flt(xg, Bg, Ag) * w.cuda()).sum().backward()
    Br, Ar, xr = Bs.clone().requires_grad_(), As.clone().requires_grad_(), x.clone().requires_grad_()
    (lti.convolve(xr, lti.iir_fsm_fir(Br, Ar, N), "causal") * w).sum().backward()
    for got, want, name in ((Bg.grad, Br.grad, "dBs"), (Ag.grad, Ar.grad, "dAs"), (xg.grad, xr.grad, "dx")):
        assert_close(got.cpu(), want, 5e-4, f"fsm order {order} {name}")
