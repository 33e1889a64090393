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
