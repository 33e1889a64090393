"""The two forms of the STFT-masked-noise impulse response synthesis behind gfx_stft_reverb_ir_sched_f32: frames as matrix
products on the fp32 matrix cores (GFX_ISTFT_GEMM) and as 192-point FFTs overlap-added in LDS (GFX_ISTFT_FFT, the
reference's default n_fft = 384 / hop = 192), against the oracle (reverb.py:161-200) and against each other."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _oracle_ir(m, p0, p1, genv, ms_lr):
    from oracle import processors as orc

    ref = orc.OracleSTFTMaskedNoiseReverb(ir_len=m.ir_len, processor_channel=m.processor_channel, n_fft=m.n_fft,
                                          hop_length=m.hop_length, gain_envelope=genv is not None)
    ir = ref.compute_ir(p0.cpu(), p1.cpu(), None if genv is None else genv.cpu())
    if ms_lr:
        ir = torch.stack([ir[:, 0] + ir[:, 1], ir[:, 0] - ir[:, 1]], 1)
    return ir


@pytest.mark.parametrize("ir_len", [60000, 3001, 5952, 5953, 400, 12000, 193, 2880, 2881])
@pytest.mark.parametrize("genv,ms_lr", [(False, True), (True, False)])
def test_fft_frames_equal_matrix_core_frames_and_the_oracle(ir_len, genv, ms_lr):
    from grafx_amd import ops
    from grafx_amd.processors import STFTMaskedNoiseReverb

    torch.manual_seed(ir_len)
    R = 5
    m = STFTMaskedNoiseReverb(ir_len=ir_len, gain_envelope=genv, flashfftconv=False).cuda()
    K, T = m.num_bins, m.num_frames
    p0 = torch.randn(R, 2, K, device="cuda")
    p1 = torch.randn(R, 2, K, device="cuda") - 3.0
    g = torch.randn(R, 2, T, device="cuda") * 0.5 if genv else None
    basis = m._istft_basis(p0.device)
    out = {}
    for sched in ("gemm", "fft", "auto"):
        out[sched] = ops.stft_reverb_ir(m.noise_stft, p0, p1, g, m.window, basis, ir_len, m.hop_length, ms_lr, schedule=sched)
    assert torch.equal(out["fft"][0], out["auto"][0]) and torch.equal(out["fft"][1], out["auto"][1])
    ref = _oracle_ir(m, p0, p1, g, ms_lr)
    scale = ref.abs().amax(dim=(1, 2), keepdim=True)
    for sched in ("gemm", "fft"):
        ir, gain = out[sched]
        assert torch.isfinite(ir).all() and torch.isfinite(gain).all()
        assert_close((ir.cpu() / scale), (ref / scale), 1e-5, f"impulse response, {sched}")
        want = 1.0 / torch.sqrt(ref.double().square().sum(-1).mean(-1) + 1e-12)
        assert_close(gain.cpu().double(), want, 1e-5, f"row gain, {sched}")
    # run to run: the same bits (fixed summation orders, no atomics)
    again = ops.stft_reverb_ir(m.noise_stft, p0, p1, g, m.window, basis, ir_len, m.hop_length, ms_lr, schedule="fft")
    assert torch.equal(again[0], out["fft"][0]) and torch.equal(again[1], out["fft"][1])


def test_fft_frames_with_fresh_noise_per_row():
    from grafx_amd import ops
    from grafx_amd.processors import STFTMaskedNoiseReverb

    torch.manual_seed(3)
    R, ir_len = 3, 7000
    m = STFTMaskedNoiseReverb(ir_len=ir_len, flashfftconv=False).cuda()
    noise = m.sample_noise(R, torch.device("cuda"))
    p0 = torch.randn(R, 2, m.num_bins, device="cuda")
    p1 = torch.randn(R, 2, m.num_bins, device="cuda") - 3.0
    basis = m._istft_basis(p0.device)
    a = ops.stft_reverb_ir(noise, p0, p1, None, m.window, basis, ir_len, m.hop_length, True, schedule="gemm")
    b = ops.stft_reverb_ir(noise, p0, p1, None, m.window, basis, ir_len, m.hop_length, True, schedule="fft")
    s = a[0].abs().max().item()
    assert (a[0] - b[0]).abs().max().item() <= 2e-6 * s
    assert_close(b[1].cpu(), a[1].cpu(), 1e-5, "row gain")


def test_fft_schedule_is_refused_where_it_does_not_apply():
    from grafx_amd import ops
    from grafx_amd.processors import STFTMaskedNoiseReverb

    m = STFTMaskedNoiseReverb(ir_len=4000, n_fft=256, hop_length=128, flashfftconv=False).cuda()
    p0 = torch.zeros(1, 2, m.num_bins, device="cuda")
    basis = m._istft_basis(p0.device)
    with pytest.raises(RuntimeError):
        ops.stft_reverb_ir(m.noise_stft, p0, p0, None, m.window, basis, 4000, 128, True, schedule="fft")
    ir, _ = ops.stft_reverb_ir(m.noise_stft, p0, p0, None, m.window, basis, 4000, 128, True, schedule="auto")
    assert torch.isfinite(ir).all()


def test_more_rows_than_one_launch_takes():
    """The kernels index rows with 16-bit grid coordinates: ops.stft_reverb_ir splits 32 780 rows into launches of 32 767
    and the rows on either side of the seam come out as they do alone (shared and per-row noise, with a gain envelope)."""
    from grafx_amd import ops
    from grafx_amd.processors import STFTMaskedNoiseReverb

    torch.manual_seed(11)
    R, ir_len = 32780, 400
    m = STFTMaskedNoiseReverb(ir_len=ir_len, gain_envelope=True, flashfftconv=False).cuda()
    p0 = torch.randn(R, 2, m.num_bins, device="cuda")
    p1 = torch.randn(R, 2, m.num_bins, device="cuda") - 3.0
    g = torch.randn(R, 2, m.num_frames, device="cuda") * 0.5
    basis = m._istft_basis(p0.device)
    ir, gain = ops.stft_reverb_ir(m.noise_stft, p0, p1, g, m.window, basis, ir_len, m.hop_length, True)
    for lo in (0, 32760, R - 10):
        a, b = ops.stft_reverb_ir(m.noise_stft, p0[lo : lo + 10], p1[lo : lo + 10], g[lo : lo + 10], m.window, basis, ir_len,
                                  m.hop_length, True)
        assert torch.equal(ir[lo : lo + 10], a) and torch.equal(gain[lo : lo + 10], b)
    noise = m.sample_noise(16, torch.device("cuda")).repeat(R // 16 + 1, 1, 1, 1)[:R].contiguous()
    ir2, gain2 = ops.stft_reverb_ir(noise, p0, p1, g, m.window, basis, ir_len, m.hop_length, False)
    sl = slice(32762, 32774)
    a, b = ops.stft_reverb_ir(noise[sl].contiguous(), p0[sl], p1[sl], g[sl], m.window, basis, ir_len, m.hop_length, False)
    assert torch.equal(ir2[sl], a) and torch.equal(gain2[sl], b)
