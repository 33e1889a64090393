"""The reference's odd-length aliasing y = irfft_{P-1}(rfft_P(z)) (core/convolution.py:123-126) on the native chirp-z
kernels (gfx_odd_alias_f32) against torch.fft in float64, for lengths around every transform-size boundary
(NFFT = C x 8192 >= (3P - 1) / 2 for C = 1 .. 32, then 4 x 32 with an outer radix-4 level), primes, and the headline length 131072 + 4000 - 1."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [3, 5, 101, 4001, 5461, 5463, 10921, 10923, 21845, 21847, 43691, 87381, 87383, 135071,
                               147455, 174761, 174763, 174765, 300001, 483999, 699051])
def test_odd_alias_matches_float64_fft(P):
    from grafx_amd import ops

    assert ops.odd_alias_supported(P) and not ops.odd_alias_supported(P + 1)
    torch.manual_seed(P)
    z = torch.randn(*((3, 2) if P < 200000 else (2, 1)), P, device="cuda")
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    got = ops.odd_alias(z)
    assert got.shape == want.shape
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= 3e-6, f"P={P}: {err:.2e}"
    lo, n = P // 3, max(1, P // 5)
    part = ops.odd_alias(z, lo, n)
    assert torch.equal(part, got[..., lo : lo + n])


def test_odd_alias_limits():
    from grafx_amd import ops

    assert not ops.odd_alias_supported(699053) and not ops.odd_alias_supported(1) and not ops.odd_alias_supported(4000)
    # structured input: a unit impulse at m0 aliases to the Dirichlet-kernel row of the resampling matrix; rows sum to 1
    P = 1001
    z = torch.zeros(1, P, device="cuda")
    z[0, 17] = 1.0
    y = ops.odd_alias(z)
    want = torch.fft.irfft(torch.fft.rfft(z.double()))
    assert (y.double() - want).abs().max() <= 1e-6
    ones = ops.odd_alias(torch.ones(2, P, device="cuda"))
    assert (ones - P / (P - 1)).abs().max() <= 1e-5   # DC gain of irfft_{P-1} o rfft_P
