"""The direct-sum real DFT pair of the parameter-side front-ends (gfx_rdft_f32 / gfx_irdft_f32) and its autograd wrapper:
against torch.fft in float64 on the CPU, odd and even lengths, and the gradient torch derives from torch.fft.irfft."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 384, 4000, 4001, 8192])
def test_rdft_matches_rfft(n):
    from grafx_amd import ops

    torch.manual_seed(n)
    x = torch.randn(3, 2, n)
    X = ops.rdft(x.cuda())
    ref = torch.fft.rfft(x.double(), dim=-1)
    assert X.shape == ref.shape and X.dtype == torch.complex64
    assert_close(torch.view_as_real(X.cpu()), torch.view_as_real(ref).float(), 2e-6, f"rdft n={n}")


@pytest.mark.parametrize("n", [2, 3, 384, 4001, 4000])
def test_small_irfft_is_differentiable_like_torch_fft_irfft(n):
    from grafx_amd import autograd as diff

    torch.manual_seed(n)
    K = n // 2 + 1
    X = torch.randn(4, K, dtype=torch.complex64)
    w = torch.randn(4, n)
    Xr = X.clone().to(torch.complex128).requires_grad_()
    yr = torch.fft.irfft(Xr, n=n, dim=-1)
    (yr * w.double()).sum().backward()
    Xg = X.cuda().requires_grad_()
    y = diff.irfft_small(Xg, n)
    (y * w.cuda()).sum().backward()
    assert_close(y.detach().cpu(), yr.detach().float(), 2e-6, f"irfft n={n}")
    assert_close(torch.view_as_real(Xg.grad.cpu()), torch.view_as_real(Xr.grad).float(), 2e-6, f"irfft gradient n={n}")


def test_training_front_ends_do_not_call_the_fft_library(monkeypatch):
    """One training step of the headline processors (ParametricEqualizer N = 4001, STFTMaskedNoiseReverb with its fixed
    noise): every transform of the parameter-side front-ends runs on the library's own kernels -- torch.fft is not
    touched, forward or backward."""
    import grafx_amd.processors as P

    def boom(*a, **k):
        raise AssertionError("torch.fft was called on the training path")

    torch.manual_seed(0)
    eq = P.ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=4001).cuda()
    rv = P.STFTMaskedNoiseReverb(ir_len=6001, flashfftconv=False).cuda()
    x = torch.randn(2, 2, 20000, device="cuda")
    pe = {k: (0.3 * torch.randn(2, 1, 6, device="cuda")).requires_grad_() for k in ("w0", "q_inv", "log_gain")}
    pr = {k: torch.randn(2, 2, 193, device="cuda").requires_grad_() for k in ("init_log_magnitude", "delta_log_magnitude")}
    for name in ("fft", "ifft", "rfft", "irfft"):
        monkeypatch.setattr(torch.fft, name, boom)
    monkeypatch.setattr(torch, "stft", boom)
    monkeypatch.setattr(torch, "istft", boom)
    y = rv(eq(x, **pe), **pr)
    y.square().mean().backward()
    for p in list(pe.values()) + list(pr.values()):
        assert p.grad is not None and torch.isfinite(p.grad).all()
