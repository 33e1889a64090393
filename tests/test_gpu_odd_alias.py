"""The reference's odd-length aliasing y = irfft_{P-1}(rfft_P(z)) (core/convolution.py:123-126) on the native chirp-z
kernels (gfx_odd_alias_f32) against torch.fft in float64, for lengths around every transform-size boundary
(NFFT = C x 8192 >= (3P - 1) / 2 for C = 1 .. 32, then one, two and three outer radix-4 levels up to 2^24 points), primes, and
the headline length 131072 + 4000 - 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [3, 5, 101, 4001, 5461, 5463, 10921, 10923, 21845, 21847, 43691, 87381, 87383, 135071,
                               147455, 174761, 174763, 174765, 300001, 349525, 349527, 483999, 699051, 699053, 1000001,
                               2796201, 2796203])
def test_odd_alias_matches_float64_fft(P):
    from grafx_amd import ops

    assert ops.odd_alias_supported(P) and not ops.odd_alias_supported(P + 1)
    torch.manual_seed(P)
    z = torch.randn(*((3, 2) if P < 200000 else (2, 1)), P, device="cuda")
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    got = ops.odd_alias(z)
    assert got.shape == want.shape
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= (3e-6 if P < 700000 else 5e-6), f"P={P}: {err:.2e}"
    lo, n = P // 3, max(1, P // 5)
    part = ops.odd_alias(z, lo, n)
    assert torch.equal(part, got[..., lo : lo + n])


def test_odd_alias_limits():
    from grafx_amd import ops

    assert ops.odd_alias_supported(11184811) and not ops.odd_alias_supported(11184813)
    assert not ops.odd_alias_supported(1) and not ops.odd_alias_supported(4000)
    # structured input: a unit impulse at m0 aliases to the Dirichlet-kernel row of the resampling matrix; rows sum to 1
    P = 1001
    z = torch.zeros(1, P, device="cuda")
    z[0, 17] = 1.0
    y = ops.odd_alias(z)
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    assert (y.double() - want).abs().max() <= 1e-6
    ones = ops.odd_alias(torch.ones(2, P, device="cuda"))
    assert (ones - P / (P - 1)).abs().max() <= 1e-5   # DC gain of irfft_{P-1} o rfft_P


@pytest.mark.parametrize("P", [3, 5, 101, 4001, 10923, 43691, 135071, 147455, 174765, 300001, 699051, 1000001, 2796203])
def test_odd_alias_adjoint_matches_float64_autograd(P):
    """gfx_odd_alias_adjoint_f32 against torch autograd of the float64 rfft / irfft pair (what the reference's backward
    computes, core/convolution.py:123-126), whole grid and a slice; and <A z, g> = <z, A^T g> on the native pair."""
    from grafx_amd import ops

    torch.manual_seed(P + 1)
    shape = (3, 2) if P < 200000 else (2, 1)
    z = torch.randn(*shape, P, device="cuda")
    Q = P - 1
    for lo, n in ((0, Q), (P // 3, max(1, P // 5))):
        g = torch.randn(*shape, n, device="cuda")
        zd = z.double().requires_grad_(True)
        torch.fft.irfft(torch.fft.rfft(zd))[..., lo : lo + n].backward(g.double())
        got = ops.odd_alias_adjoint(g, P, lo)
        assert got.shape == z.shape
        err = (got.double() - zd.grad).abs().max() / zd.grad.abs().max()
        assert err <= (3e-6 if P < 700000 else 5e-6), f"P={P} lo={lo}: {err:.2e}"
        lhs = (ops.odd_alias(z, lo, n).double() * g.double()).sum()
        rhs = (z.double() * got.double()).sum()
        assert abs(lhs - rhs) <= 1e-5 * (z.double().norm() * g.double().norm()), f"P={P}: {lhs} vs {rhs}"


def test_convolve_gradient_with_aliasing_is_native():
    """autograd.convolve at an odd P: gradients equal torch autograd of the reference's expression in float64, and no
    float64 tensor is created on the way (the aliasing step's backward is OddAliasFn)."""
    from grafx_amd import autograd as diff

    torch.manual_seed(5)
    L, N = 8192, 1000      # P = 9191
    x = torch.randn(3, 2, L, device="cuda", requires_grad=True)
    h = (torch.randn(3, 2, N, device="cuda") / 30).requires_grad_(True)
    g = torch.randn(3, 2, L, device="cuda")
    for mode, lo in (("causal", 0), ("zerophase", N // 2)):
        x.grad = h.grad = None
        y = diff.convolve(x, h, mode)
        assert y.dtype == torch.float32 and type(y.grad_fn).__name__ == "OddAliasFnBackward"
        y.backward(g)
        xd, hd = x.detach().double().requires_grad_(True), h.detach().double().requires_grad_(True)
        P = L + N - 1
        yd = torch.fft.irfft(torch.fft.rfft(xd, n=P) * torch.fft.rfft(hd, n=P))[..., lo : lo + L]
        yd.backward(g.double())
        assert (y.double() - yd).abs().max() <= 1e-5 * yd.abs().max()
        for a, b in ((x.grad, xd.grad), (h.grad, hd.grad)):
            assert (a.double() - b).abs().max() <= 2e-5 * b.abs().max()


@pytest.mark.parametrize("P", [3, 101, 4001, 10923, 87383, 147455, 174765, 300001, 1000001, 2796203])
def test_precise_odd_alias_is_float64_accurate(P):
    """gfx_odd_alias_precise_f32 (double-precision transforms, fp32 in / out): the result is the float64 FFT's rounded to
    fp32 -- also where the signal is 1e-6 of its peak, which is what the energy envelope needs -- and so is the adjoint."""
    from grafx_amd import ops

    torch.manual_seed(P + 2)
    shape = (3, 2) if P < 200000 else (2, 1)
    z = torch.randn(*shape, P, device="cuda")
    z[..., P // 2 :] *= 1e-6                       # a quiet passage next to a loud one
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    got = ops.odd_alias(z, precise=True)
    assert got.dtype == torch.float32
    tol = 1.5e-7 * want.abs() + 1e-12 * want.abs().max()
    assert ((got.double() - want).abs() <= tol).all(), f"P={P}: {((got.double() - want).abs() / want.abs().max()).max():.2e}"
    if P > 4000:   # the fp32 transforms' noise floor on the quiet passage, which the double-precision ones do not have
        q = slice(P - P // 4, None)
        plain = ops.odd_alias(z)
        assert (plain.double() - want)[..., q].abs().max() > 100 * (got.double() - want)[..., q].abs().max()
    lo, n = P // 3, max(1, P // 5)
    assert torch.equal(ops.odd_alias(z, lo, n, precise=True), got[..., lo : lo + n])
    g = torch.randn(*shape, n, device="cuda")
    zd = z.double().requires_grad_(True)
    torch.fft.irfft(torch.fft.rfft(zd))[..., lo : lo + n].backward(g.double())
    adj = ops.odd_alias_adjoint(g, P, lo, precise=True)
    assert (adj.double() - zd.grad).abs().max() <= 2e-7 * zd.grad.abs().max()


def test_longer_than_the_kernels_reach_fails_loudly():
    from grafx_amd.processors.core.convolution import odd_length_alias

    with pytest.raises(NotImplementedError):
        odd_length_alias(torch.zeros(1, 11184813, device="cuda"))


def test_three_outer_levels():
    """P = 3,000,001: a 2^24-point transform (64 sub-transforms of 32 x 8192 under three radix-4 levels), one row."""
    from grafx_amd import ops

    P = 3000001
    torch.manual_seed(P)
    z = torch.randn(1, P, device="cuda")
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    got = ops.odd_alias(z)
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= 6e-6, f"{err:.2e}"
    g = torch.randn(1, 1000, device="cuda")
    zd = z.double().requires_grad_(True)
    torch.fft.irfft(torch.fft.rfft(zd))[..., 5000:6000].backward(g.double())
    adj = ops.odd_alias_adjoint(g, P, 5000)
    assert (adj.double() - zd.grad).abs().max() <= 6e-6 * zd.grad.abs().max()
    ops._ALIAS_PLANS.clear()   # half a gigabyte of plan


_TILE_COUNTS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16, 18, 20, 21, 24, 25, 27, 28, 30, 32]


@pytest.mark.parametrize("C", _TILE_COUNTS)
def test_every_supported_tile_count(C):
    """The transform runs on C x 8192 points with C the smallest tile count of prime factors up to 7 that covers
    (3P - 1) / 2 (csrc/small_dft.hpp): one P per supported C -- the largest odd length that still fits C tiles -- forward in
    both precisions and the adjoint, against torch.fft in float64."""
    from grafx_amd import ops

    P = (2 * C * 8192 + 1) // 3
    P -= 1 - (P & 1)                                   # odd, and (3P - 1) / 2 <= C * 8192
    assert ops.lib().gfx_odd_alias_workspace_bytes(1, P) == C * 8192 * 8, "the geometry did not pick this tile count"
    torch.manual_seed(C)
    z = torch.randn(2, P, device="cuda")
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    for precise, tol in ((False, 3e-6), (True, 2e-7)):
        got = ops.odd_alias(z, precise=precise)
        err = (got.double() - want).abs().max() / want.abs().max()
        assert err <= tol, f"C={C} P={P} precise={precise}: {err:.2e}"
    g = torch.randn(2, P - 1, device="cuda")
    gz = ops.odd_alias_adjoint(g, P)
    lhs = (ops.odd_alias(z).double() * g.double()).sum()
    rhs = (z.double() * gz.double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * (z.double().norm() * g.double().norm()), f"C={C}: <Az, g> {lhs} vs <z, A'g> {rhs}"


@pytest.mark.parametrize("P,C", [(4607, 1), (135071, 2), (191071, 2)])
def test_odd_alias_writes_strided_buffer_rows_in_place(P, C):
    """gfx_odd_alias_rows_f32: the aliased rows land directly in a strided (B, n, C, length) view of a signal buffer (the
    stage's output slice in render_grafx) -- same values as the contiguous result, nothing outside the view touched,
    across the chunk boundary (more rows than one launch chain takes)."""
    from grafx_amd import ops

    torch.manual_seed(P)
    B, n = 3, 2
    lo, length = 5, P - 1 - 9
    z = torch.randn(B * n * C, P, device="cuda")
    want = ops.odd_alias(z, lo, length)
    buf = torch.full((B, n + 3, C, length + 7), float("nan"), device="cuda")
    view = buf[:, 2 : 2 + n, :, :length]
    got = ops.odd_alias(z, lo, length, out=view, rows_per_chunk=5)
    assert got is view
    assert torch.equal(view.reshape(B * n * C, length), want)
    mask = torch.ones_like(buf, dtype=torch.bool)
    mask[:, 2 : 2 + n, :, :length] = False
    assert torch.isnan(buf[mask]).all()
