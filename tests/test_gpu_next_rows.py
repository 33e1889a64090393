"""GPU parity of the "next row" processors against the reference's own outputs."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-5


def run(m, x, **p):
    with torch.no_grad():
        out = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()})
    return out


@pytest.mark.parametrize("name", ["LowPassFilter", "HighPassFilter", "BandPassFilter", "BandRejectFilter", "AllPassFilter"])
@pytest.mark.parametrize("N", [256, 257])
def test_parametric_filters(golden, name, N):
    import grafx_amd.processors as P

    g = golden("g10_next_rows")
    m = getattr(P, name)(flashfftconv=False, fsm_fir_len=N)
    p = {k: g[f"{k}_{name}_N{N}"] for k in m.parameter_size()}
    assert_close(run(m, g["x"], **p).cpu(), g[f"y_{name}_N{N}"], 3e-5, name)


@pytest.mark.parametrize("name", ["PeakingFilter", "LowShelf", "HighShelf"])
def test_equaliser_sections(golden, name):
    import grafx_amd.processors as P

    g = golden("g10_next_rows")
    m = getattr(P, name)(num_filters=2, flashfftconv=False, fsm_fir_len=257)
    p = {k: g[f"{k}_{name}"] for k in m.parameter_size()}
    assert_close(run(m, g["x"], **p).cpu(), g[f"y_{name}"], 3e-5, name)


def test_state_variable_filter(golden):
    import grafx_amd.processors as P

    g = golden("g10_next_rows")
    m = P.StateVariableFilter(num_filters=2, flashfftconv=False, fsm_fir_len=257)
    p = {k: g[f"{k}_svf"] for k in m.parameter_size()}
    assert_close(run(m, g["x"], **p).cpu(), g["y_svf"], 3e-5, "svf")


@pytest.mark.parametrize("L", [1024, 1023])
def test_zero_phase_fir_equalizer(golden, L):
    import grafx_amd.processors as P

    g = golden("g10_next_rows")
    m = P.ZeroPhaseFIREqualizer(num_magnitude_bins=128)
    y = run(m, g[f"x_zpfir_L{L}"], log_magnitude=g[f"lm_zpfir_L{L}"])
    assert_close(y.cpu(), g[f"y_zpfir_L{L}"], TOL, "zero-phase FIR eq")


@pytest.mark.parametrize("name,kw", [("ApproxCompressor", "iir_len"), ("ApproxNoiseGate", "freq_sample_n")])
def test_approx_dynamics(golden, name, kw):
    import grafx_amd.processors as P

    g = golden("g10_next_rows")
    m = getattr(P, name)(**{kw: 255}, flashfftconv=False)
    p = {k: g[f"{k}_{name}"] for k in m.parameter_size()}
    assert_close(run(m, g["x"], **p).cpu(), g[f"y_{name}"], 2e-5, name)


def test_containers_and_stereo_utils(golden):
    import grafx_amd.processors as P

    g = golden("g10_next_rows")
    x = g["x"]
    gain, lp = P.StereoGain(), P.LowPassFilter(flashfftconv=False, fsm_fir_len=257)
    pg = {"log_gain": g["dw_lg"].cuda()}
    pl = {"w0": g["lp_w0"].cuda(), "q_inv": g["lp_q_inv"].cuda()}
    with torch.no_grad():
        xc = x.cuda()
        assert_close(P.DryWet(gain)(xc, drywet_weight=g["dw_w"].cuda(), **pg).cpu(), g["y_drywet"], TOL, "drywet")
        y, _ = P.SerialChain({"g": gain, "lp": lp}).cuda()(xc, g=pg, lp=pl)
        assert_close(y.cpu(), g["y_serial"], 3e-5, "serial")
        y, _ = P.ParallelMix({"g": gain, "lp": lp}).cuda()(xc, parallel_weights=g["pm_w"].cuda(), g=pg, lp=pl)
        assert_close(y.cpu(), g["y_parallel"], 3e-5, "parallel")
        y, inter = P.GainStagingRegularization(gain)(xc, **pg)
        assert_close(y.cpu(), g["y_gsr"], TOL, "gsr y")
        assert abs(float(inter["gain_reg"]) - float(g["gsr_reg"])) < 1e-4
        assert_close(P.SideGainImager()(xc, g["side_lg"].cuda()).cpu(), g["y_side"], 1e-6, "side gain")
        mid, side = P.StereoToMidSide()(xc)
        assert_close(mid.cpu(), g["ms_mid"], 1e-6, "mid")
        assert_close(P.MidSideToStereo()(mid, side).cpu(), g["y_ms2lr"], 1e-6, "ms->lr")
        assert P.MonoToStereo()(xc[:, :1]).shape == (3, 2, 1023)
