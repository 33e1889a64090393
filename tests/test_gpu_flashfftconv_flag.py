"""`flashfftconv=True` (upstream's constructor default) selects, upstream, FlashFFTConv: the plain causal
convolution (reference core/convolution.py:85-106).  Here it selects the same mathematical operation on the fp32
HIP overlap-save kernels; `flashfftconv=False` keeps upstream's native torch.fft semantics including the odd-length
aliasing.  The two differ only when L + N - 1 is odd, which is the case for every upstream default length at
L = 131072."""
import pytest
import torch

from conftest import assert_close, rel_err

pytestmark = pytest.mark.gpu


def test_fir_convolution_flag_switches_between_plain_and_aliased_convolution():
    from grafx_amd.processors import FIRConvolution
    from oracle import lti

    torch.manual_seed(0)
    x, h = torch.randn(3, 2, 4096), torch.randn(3, 1, 512) / 16   # P = 4607, odd
    with torch.no_grad():
        y_flash = FIRConvolution(mode="causal", flashfftconv=True)(x.cuda(), h.cuda()).cpu()
        y_native = FIRConvolution(mode="causal", flashfftconv=False)(x.cuda(), h.cuda()).cpu()
    assert_close(y_flash, lti.linear_convolve(x, h, "causal"), 1e-5, "flashfftconv=True: plain causal convolution")
    assert_close(y_native, lti.convolve(x, h, "causal"), 1e-5, "flashfftconv=False: upstream's native path")
    assert rel_err(y_flash, y_native)[0] > 1e-3
    with pytest.raises(AssertionError):  # upstream: no zero-phase mode with FlashFFTConv (convolution.py:86-89)
        with pytest.warns(UserWarning):
            conv = FIRConvolution(mode="zerophase", flashfftconv=True)
        conv(x.cuda(), h.cuda())


def test_default_constructor_arguments_run_plain_convolutions_end_to_end():
    """ParametricEqualizer(), Compressor(), STFTMaskedNoiseReverb() with upstream's defaults (4000 / 16384 / 60000
    taps, flashfftconv=True) against the oracle evaluated with true linear convolutions."""
    import grafx_amd.processors as P
    import oracle
    from oracle import lti

    torch.manual_seed(1)
    L = 8192
    x = torch.randn(2, 2, L)
    import oracle.processors as oproc

    saved = lti.convolve

    def use(fn):  # the oracle's processors bound `convolve` by name; its smoothers look it up in lti
        lti.convolve = fn
        oproc.convolve = fn
    cases = [
        (P.ParametricEqualizer(num_filters=6), oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=4000),
         {k: 0.3 * torch.randn(2, 1, 6) for k in ("w0", "q_inv", "log_gain")}, 2e-5),
        (P.Compressor(energy_smoother="iir"), oracle.OracleCompressor(energy_smoother="iir", iir_len=16384),
         {"log_threshold": torch.randn(2, 1) - 2, "log_ratio": torch.randn(2, 1), "log_knee": torch.randn(2, 1),
          "z_alpha_pre": torch.randn(2, 1) + 2}, 5e-5),
        (P.STFTMaskedNoiseReverb(), oracle.OracleSTFTMaskedNoiseReverb(ir_len=60000),
         {"init_log_magnitude": torch.randn(2, 2, 193), "delta_log_magnitude": torch.randn(2, 2, 193)}, 2e-5),
    ]
    try:
        for m, o, p, tol in cases:
            with torch.no_grad():
                y = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
                use(saved)
                y_alias = o(x, **p)
                use(lti.linear_convolve)                     # the oracle with plain convolutions
                y_plain = o(x, **p)
            assert_close(y, y_plain, tol, type(m).__name__ + " (flashfftconv=True)")
            assert rel_err(y_plain, y_alias)[0] > 1e-4, "the two semantics must differ here (odd L + N - 1)"
    finally:
        use(saved)
