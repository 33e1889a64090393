"""Short-FIR Toeplitz-GEMM kernel on the fp32 matrix cores (gfx_fir_direct_f32) against the oracle's linear
convolution and against the FFT tile kernel: tap counts around the 16-tap block boundaries, signal lengths around the
4096-sample segment and 256-sample tile boundaries, output windows / offsets, channel broadcasts, shared filters and
strided buffer views."""
import random

import pytest
import torch

from oracle import lti

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(48))
def test_fir_direct_matches_the_oracle_convolution(seed):
    from grafx_amd import ops

    rng = random.Random(500 + seed)
    torch.manual_seed(seed)
    L = rng.choice([1, 2, 15, 16, 17, 255, 256, 257, 1000, 4095, 4096, 4097, 8193, 20000, 70001])
    N = rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 400, 511, 512])
    Cin, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    B, n = rng.choice([(1, 1), (2, 3), (3, 2)])
    shared = rng.random() < 0.5
    full = L + N - 1
    off = rng.choice([0, 0, N // 2, N - 1, rng.randint(0, N - 1)])
    Lout = max(1, min(rng.choice([L, full - off, max(1, min(L, 777))]), full - off))
    buf = torch.randn(B, n + 2, Cin, L, device="cuda")
    x4 = buf.narrow(1, 1, n)                                  # strided (B, n, C, L) view
    h = torch.randn(n if shared else B * n, Cf, N, device="cuda") / max(N, 1) ** 0.5
    y = ops.fir_direct(x4, h, Lout=Lout, off=off)
    hx = (h.repeat(B, 1, 1) if shared else h).cpu()
    ref = lti.linear_convolve(x4.reshape(B * n, Cin, L).cpu(), hx, "full")[..., off : off + Lout]
    what = f"L={L} N={N} C={Cin}/{Cf} B={B} n={n} off={off} Lout={Lout} shared={shared}"
    assert y.shape == ref.shape, what
    scale = ref.abs().max().clamp_min(1e-6)
    assert torch.isfinite(y).all(), what
    assert (y.cpu() - ref).abs().max() <= 1e-5 * scale, what
    y_fft = ops.fftconv(x4, ops.fir_spectrum(h.reshape(-1, N)), N, Cf, Lout=Lout, off=off, h_rows=h.shape[0])
    assert (y - y_fft).abs().max() <= 4e-6 * scale, what


def test_fir_direct_writes_into_a_buffer_view_and_rejects_long_filters():
    from grafx_amd import ops
    from grafx_amd._lib import GfxError

    torch.manual_seed(0)
    B, n, L, N = 2, 3, 5000, 33
    x4 = torch.randn(B, n, 2, L, device="cuda")
    h = torch.randn(n, 1, N, device="cuda")
    buf = torch.zeros(B, n + 2, 2, L, device="cuda")
    ops.fir_direct(x4, h, out=buf.narrow(1, 1, n))
    want = ops.fir_direct(x4, h)
    assert torch.equal(buf[:, 1 : n + 1].reshape(B * n, 2, L), want)
    assert (buf[:, 0] == 0).all() and (buf[:, n + 1] == 0).all()
    with pytest.raises(GfxError):
        ops.fir_direct(x4, torch.randn(n, 1, 513, device="cuda"))


def test_convolve_routes_short_filters_to_the_matrix_cores():
    """convolve() (the FIRConvolution entry) with a short filter gives the same result as the FFT tile path."""
    from grafx_amd import ops
    from grafx_amd.processors.core.convolution import SHORT_FIR_TAPS, convolve

    torch.manual_seed(1)
    x = torch.randn(4, 2, 30000, device="cuda")
    for N in (9, SHORT_FIR_TAPS - (SHORT_FIR_TAPS % 2 == 0), 2 * SHORT_FIR_TAPS + 1):   # odd N: even L + N - 1
        h = torch.randn(4, 1, N, device="cuda") / N ** 0.5
        for mode in ("causal", "zerophase"):
            y = convolve(x, h, mode=mode)
            off = 0 if mode == "causal" else N // 2
            want = ops.fftconv(x, ops.fir_spectrum(h.reshape(-1, N)), N, 1, Lout=x.shape[-1], off=off)
            assert (y - want).abs().max() <= 4e-6 * want.abs().max(), (N, mode)
