"""GPU parity on ragged / tiny / awkward shapes (edge cases of the hot path)."""
import pytest
import torch

import oracle
from conftest import assert_close, assert_parity
from oracle import lti

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,N", [(1, 2), (1, 4), (2, 3), (3, 2), (5, 8), (17, 17), (100, 4097), (16385, 16385),
                                  (12384, 4001), (12385, 4001), (24768, 4001), (8192, 8194), (33, 60001)])
def test_convolve_ragged_shapes(L, N):
    from grafx_amd.processors import convolve

    torch.manual_seed(L * 7 + N)
    x, h = torch.randn(2, 2, L), torch.randn(2, 1, N) / N**0.5
    for mode in ("causal", "zerophase"):
        with torch.no_grad():
            y = convolve(x.cuda(), h.cuda(), mode=mode).cpu()
        ref = oracle.convolve(x.double(), h.double(), mode).float()  # reference semantics (both parities)
        assert y.shape == ref.shape
        assert_close(y, ref, 2e-5, f"convolve L={L} N={N} {mode}")


@pytest.mark.parametrize("L", [1, 3, 4, 5, 1023, 1024, 1025, 4099])
@pytest.mark.parametrize("C", [1, 2])
def test_compressor_ragged_lengths(L, C):
    from grafx_amd.processors import Compressor

    torch.manual_seed(L + C)
    x = torch.randn(3, C, L)
    p = {"log_threshold": torch.randn(3, 1), "log_ratio": torch.randn(3, 1), "log_knee": torch.randn(3, 1),
         "z_alpha_pre": torch.randn(3, 1)}
    for iir_len in (63, 64):
        m = Compressor(energy_smoother="iir", iir_len=iir_len, flashfftconv=False)
        with torch.no_grad():
            y = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        ref64 = oracle.OracleCompressor(iir_len=iir_len)(x.double(), **{k: v.double() for k, v in p.items()}).float()
        ref32 = oracle.OracleCompressor(iir_len=iir_len)(x, **p)
        assert_parity(y, ref32, ref64, 2e-5, f"compressor L={L} C={C} iir_len={iir_len}")


def test_degenerate_length_fails_like_the_reference():
    """L = N = 1 gives P = 1: the reference's irfft has nothing to invert and raises; so do we."""
    from grafx_amd.processors import convolve

    x, h = torch.randn(1, 1, 1), torch.randn(1, 1, 1)
    with pytest.raises(RuntimeError):
        oracle.convolve(x, h, "causal")
    with pytest.raises(RuntimeError):
        convolve(x.cuda(), h.cuda(), mode="causal")


@pytest.mark.parametrize("R,L", [(1, 7), (65, 130), (3, 1), (130, 63)])
def test_ballistics_ragged(R, L):
    from grafx_amd.processors import Ballistics

    torch.manual_seed(R + L)
    u, z = torch.rand(R, L) * 2, torch.randn(R, 2)
    with torch.no_grad():
        y = Ballistics()(u.cuda(), z.cuda()).cpu()
    assert_close(y, oracle.ballistics(u, z), 1e-5, "ballistics")


def test_single_row_mono_eq_and_gain():
    from grafx_amd.processors import ParametricEqualizer, StereoGain

    torch.manual_seed(1)
    x = torch.randn(1, 1, 777)
    p = {k: torch.randn(1, 1, 3) for k in ("w0", "q_inv", "log_gain")}
    m = ParametricEqualizer(num_filters=3, flashfftconv=False, fsm_fir_len=65)
    with torch.no_grad():
        y = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        g = StereoGain()(x.cuda(), torch.randn(1, 2).cuda())
    assert_close(y, oracle.OracleParametricEqualizer(num_filters=3, fsm_fir_len=65)(x, **p), 2e-5, "mono eq")
    assert g.shape == (1, 2, 777)  # mono input broadcasts to stereo, as upstream


def test_fsm_fir_lengths_and_limits():
    from grafx_amd.processors import IIRFilter

    torch.manual_seed(2)
    Bs = torch.randn(2, 1, 2, 3) * 0.2 + torch.tensor([1.0, 0, 0])
    As = torch.tensor([1.0, -1.2, 0.5]).expand(2, 1, 2, 3).contiguous()
    for N in (2, 3, 64, 1000, 4095, 4096, 4097, 8192):   # the last two: beyond the native tile, torch front-end
        f = IIRFilter(flashfftconv=False, fsm_fir_len=N)
        with torch.no_grad():
            fir = f.fsm_fir(Bs.cuda(), As.cuda()).cpu()
        assert_close(fir, lti.iir_fsm_fir(Bs, As, N), 1e-5, f"fsm N={N}")


def test_render_3d_input_and_index_reads():
    """3-D (unbatched) input and a graph whose reads are index-based (no reordering applied)."""
    from grafx_amd.data import GRAFX, NodeConfigs, convert_to_tensor
    from grafx_amd.processors import StereoGain
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render

    G = GRAFX(config=NodeConfigs(["gain"]))
    ins = [G.add("in") for _ in range(3)]
    gains = [G.add("gain") for _ in range(3)]
    out = G.add("out")
    for i, g in zip(ins, gains[::-1]):  # crossed wiring -> index reads
        G.connect(i, g)
    for g in gains:
        G.connect(g, out)
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    torch.manual_seed(3)
    x = torch.randn(3, 2, 500)
    lg = torch.randn(3, 2)
    want, _, _ = render_grafx({"gain": oracle.OracleStereoGain()}, x, {"gain": {"log_gain": lg}}, rd)
    rd_gpu = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")
    with torch.no_grad():
        got, _, buf = render_grafx({"gain": StereoGain()}, x.cuda(), {"gain": {"log_gain": lg.cuda()}}, rd_gpu)
    assert got.shape == want.shape
    assert_close(got.cpu(), want, 1e-6, "3-D render")


@pytest.mark.gpu
@pytest.mark.parametrize("N", [8192, 6001, 16384])
def test_fsm_fir_len_beyond_the_native_tile(N):
    """fsm_fir_len > 4096: the taps come from the torch front-end (same formula, float64 inverse FFT), the
    convolution from the HIP kernels (single tile up to 8193 taps, partitioned beyond)."""
    import torch

    import grafx_amd.processors as P
    import oracle

    torch.manual_seed(0)
    L = 20001 if N % 2 == 0 else 20000   # keep L + N - 1 even: the reference's own exact case
    x = torch.randn(2, 2, L)
    p = {k: 0.3 * torch.randn(2, 1, 4) for k in ("w0", "q_inv", "log_gain")}
    m = P.ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=N).cuda()
    o = oracle.OracleParametricEqualizer(num_filters=4, fsm_fir_len=N)
    with torch.no_grad():
        y = m(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
    assert_close(y, o(x, **p), 2e-5, f"PEQ fsm_fir_len={N}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_DYN_SEEDS", 64))))
def test_random_dynamics_configurations_match_the_oracle(seed):
    """Compressor / NoiseGate over random knees, smoothers, lengths (ragged, shorter and longer than the smoother,
    both parities), channel counts and row counts (time-chunked and serial scans) against the CPU oracle."""
    import random

    import torch

    import grafx_amd.processors as P
    import oracle

    rng = random.Random(seed)
    torch.manual_seed(seed)
    gate = rng.random() < 0.5
    knee = rng.choice(["hard", "quadratic", "exponential"])
    smoother = rng.choice(["iir", "iir", None, "ballistics"])
    iir_len = rng.choice([1, 2, 63, 1023, 1024, 16383])
    L = rng.choice([1, 5, 1023, 1024, 1025, 4099, 20000, 40001])
    R, C = rng.choice([(1, 2), (3, 1), (5, 2), (70, 2)])
    cls, ocls = (P.NoiseGate, oracle.OracleNoiseGate) if gate else (P.Compressor, oracle.OracleCompressor)
    m = cls(energy_smoother=smoother, knee=knee, iir_len=iir_len, flashfftconv=False).cuda()
    o = ocls(energy_smoother=smoother, knee=knee, iir_len=iir_len)
    x = torch.randn(R, C, L) * torch.rand(R, 1, 1)
    p = {"log_threshold": torch.randn(R, 1) - 2, "log_ratio": torch.randn(R, 1)}
    if knee != "hard":
        p["log_knee"] = torch.randn(R, 1)
    if smoother == "iir":
        p["z_alpha_pre"] = torch.randn(R, 1) * 3
    elif smoother == "ballistics":
        p["z_alpha_pre"] = torch.randn(R, 2)
    what = f"gate={gate} knee={knee} smoother={smoother} N={iir_len} L={L} R={R} C={C}"
    with torch.no_grad():
        try:
            ref32 = o(x, **p)
        except RuntimeError:
            # odd L + N - 1 with a smoother so short that upstream's aliased convolution returns fewer than L samples:
            # the reference fails on the shape mismatch, and so must we (not read past the end of a buffer)
            with pytest.raises((RuntimeError, ValueError)):
                m(x.cuda(), **{k: v.cuda() for k, v in p.items()})
            return
        y = m(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        ref64 = o(x.double(), **{k: v.double() for k, v in p.items()}).float()
    assert_parity(y, ref32, ref64, 2e-5, what)


@pytest.mark.gpu
def test_single_band_equalizer_with_shelving_fails_like_upstream():
    """use_shelving_filters splits the bands [1, K-2, 1] upstream (eq.py:254, 300-302): K = 1 cannot be split and
    torch.split raises; so do we, from forward() and from the render's ahead-of-time prepare()."""
    import grafx_amd.processors as P

    m = P.ParametricEqualizer(num_filters=1, flashfftconv=False, fsm_fir_len=257).cuda()
    p = {k: torch.zeros(2, 1, 1, device="cuda") for k in ("w0", "q_inv", "log_gain")}
    with pytest.raises(RuntimeError):
        m(torch.randn(2, 2, 1000, device="cuda"), **p)
    with pytest.raises(RuntimeError):
        m.prepare(**p)
    ok = P.ParametricEqualizer(num_filters=1, use_shelving_filters=False, flashfftconv=False, fsm_fir_len=257).cuda()
    assert ok(torch.randn(2, 2, 1000, device="cuda"), **p).shape == (2, 2, 1000)
