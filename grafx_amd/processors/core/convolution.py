"""FIR convolution backend (mirrors grafx.processors.core.convolution —
reference src/grafx/processors/core/convolution.py:17-134) on the HIP overlap-save kernels.

Semantics are the reference's, including its length-parity quirk: `convolve`
pads to P = Lx + Lh - 1, multiplies rffts and calls ``irfft`` *without* ``n``, so
for odd P it inverts a P-point spectrum on a (P-1)-point grid
(convolution.py:120-126).  Here

* even P  -> the result IS the linear convolution: one pass of the HIP
  overlap-save kernels (gfx_fftconv_f32), reading x once and writing y once;
* odd P   -> the HIP kernels produce the full linear convolution z (length P) and
  the reference's aliasing ``irfft_{P-1}(rfft_P(z))`` is applied on top, natively as two
  chirp-z transforms on the LDS FFT tile (gfx_odd_alias_f32, its transpose for the gradient,
  double-precision transforms for the dynamics envelope; no FFT library; P <= 11,184,811,
  longer signals raise -- DESIGN.md §2).

``flashfftconv=True`` -- the constructor default of every upstream processor -- is resolved the way the reference's
PyTorch-CPU path resolves it (convolution.py:47-51): FlashFFTConv is a CUDA library, so the flag falls back to the
native convolve() above with upstream's warning, aliasing included (pinned by golden g17).

``set_exact_convolution(True)`` is the explicit opt-out of the quirk (true linear convolution
for every length; deviates from the reference when P is odd).
"""
import contextlib
import contextvars
import warnings

import torch
import torch.nn as nn

from ... import autograd as diff
from ... import ops
from ...autograd import needs_grad

_EXACT = contextvars.ContextVar("grafx_amd_exact_convolution", default=False)   # context-local, see set_exact_convolution
# Filters up to this many taps run as Toeplitz GEMMs on the fp32 matrix cores (gfx_fir_direct_f32) instead of the FFT tile
# kernel: the crossover measured on MI355X (profiles/r2/fir_mfma_crossover.txt).
SHORT_FIR_TAPS = 40


def set_exact_convolution(flag=True):
    """Force true linear convolution even where the reference aliases (odd Lx+Lh-1).  The setting is a ContextVar: it
    holds for the calling thread / async task (and contexts copied from it), not for unrelated threads."""
    _EXACT.set(bool(flag))


def exact_convolution():
    return _EXACT.get()


@contextlib.contextmanager
def exact_convolution_scope(flag):
    """Run a block under a given setting whatever thread it runs on: the render's autograd node records the caller's
    setting in forward and re-traces its stages under it in backward, which the autograd engine runs on its own worker
    thread (a thread that does not inherit the caller's context)."""
    token = _EXACT.set(bool(flag))
    try:
        yield
    finally:
        _EXACT.reset(token)


FLASHFFTCONV_AVAILABLE = False   # a CUDA library (convolution.py:9-14): never importable next to a HIP device


def resolve_flashfftconv(flashfftconv, warn=True):
    """What upstream's FIRConvolution.__init__ makes of the ``flashfftconv`` constructor argument when FlashFFTConv
    cannot be imported -- always the case on the reference's PyTorch-CPU path, the parity target (convolution.py:47-53):
    a warning and the native convolve().  Returns the resolved flag (False).  ``warn=False`` for a wrapper whose inner
    module has already warned (upstream warns once per FIRConvolution it builds)."""
    if flashfftconv and warn:
        warnings.warn("FlashFFTConv is not available. Using native convolution instead.")
    return False


def reference_aliases(lx, lh, exact=False):
    """True when the reference's convolve() is NOT a linear convolution for these lengths (odd lx + lh - 1,
    convolution.py:119-134).  ``exact`` is a call site's own request for the true linear convolution (the processors
    pass their resolved ``flashfftconv`` attribute, which is False: see resolve_flashfftconv)."""
    return (lx + lh - 1) % 2 == 1 and not (_EXACT.get() or exact)


def odd_length_alias(z, lo=0, length=None, precise=False):
    """irfft_{P-1}(rfft_P(z))[..., lo : lo + length] for a full linear convolution z of odd length P
    (convolution.py:123-126): two chirp-z transforms on the LDS FFT tile (gfx_odd_alias_f32, no FFT library).

    The gradient is the transposed pair of transforms (autograd.OddAliasFn).  ``precise=True`` carries the transforms
    in double precision (gfx_odd_alias_precise_f32) -- for the energy-envelope smoother of the dynamics processors,
    whose output feeds log() and a gain curve: there the fp32 chirp-z noise (~1e-6 of the peak, about twice what the
    reference's own mixed-radix fp32 FFT leaves) is amplified on quiet passages beyond the parity bound
    (tests/test_gpu_edge_cases.py::test_compressor_ragged_lengths).  P <= 11,184,811 (a 2^24-point transform must cover
    1.5 P: 233 s of audio at 48 kHz); longer signals raise."""
    P = z.shape[-1]
    length = P - 1 - lo if length is None else length
    if not ops.odd_alias_supported(P):
        raise NotImplementedError(f"convolve: the reference's odd-length aliasing (P = Lx + Lh - 1 = {P}) is implemented for "
                                  "P <= 11,184,811; use a filter length that makes P even, or "
                                  "set_exact_convolution(True) for the plain linear convolution")
    if torch.is_grad_enabled() and z.requires_grad:
        from ... import autograd as diff

        return diff.odd_alias(z, lo, length, precise)
    return ops.odd_alias(z, lo, length, precise=precise)


def compute_pad_len(x, y, pad_mode="min"):
    if pad_mode != "min":
        # the reference's "pow2" branch never returns (convolution.py:111-114), so it cannot be relied on
        raise ValueError(f"pad_mode={pad_mode!r} is not supported (only 'min', the reference default)")
    return x.shape[-1] + y.shape[-1] - 1


def convolve_taps(x, Hs, N, Cf, mode, out=None, tee=None, exact=False, h_rows=None):
    """convolve() given precomputed tile spectra of the taps.

    ``x`` is (R,C,L) or a strided (B,n,C,L) view of the signal buffer; with ``out`` (same kind of
    view) the kernels write the result in place and ``out`` is returned.  ``tee`` (a view shaped like
    ``x``) additionally receives a copy of ``x`` -- from the convolution kernel itself when it can."""
    L = x.shape[-1]
    if tee is not None:
        if mode == "causal" and not reference_aliases(L, N, exact) and ops.fftconv_can_tee(x.shape[-2], Cf, L, L, 0, N):
            return ops.fftconv(x, Hs, N, Cf, Lout=L, off=0, out=out, tee=tee, h_rows=h_rows)
        if reference_aliases(L, N, exact) and ops.fftconv_can_tee(x.shape[-2], Cf, L, L + N - 1, 0, N):
            pass        # the full-length convolution below writes the input through
        else:
            tee.copy_(x)
            tee = None
    if not reference_aliases(L, N, exact):
        if mode == "causal":
            return ops.fftconv(x, Hs, N, Cf, Lout=L, off=0, out=out, h_rows=h_rows)
        if mode == "zerophase":
            return ops.fftconv(x, Hs, N, Cf, Lout=L, off=N // 2, out=out, h_rows=h_rows)
        return ops.fftconv(x, Hs, N, Cf, Lout=L + N - 1, off=0, out=out, h_rows=h_rows)
    lo, length = {"causal": (0, L), "zerophase": (N // 2, L)}.get(mode, (0, L + N - 2))
    rm = {}     # the rows' maxima, when the convolution kernel leaves them (the pair scaling of the aliasing: ops.odd_alias)
    z = ops.fftconv(x, Hs, N, Cf, Lout=L + N - 1, off=0, h_rows=h_rows, tee=tee, rowmax=rm)
    if out is not None and not (torch.is_grad_enabled() and z.requires_grad):
        # the last chirp-z pass writes the rows of the buffer view in place
        return ops.odd_alias(z, lo, length, out=out, rowmax=rm.get("words"))
    if not (torch.is_grad_enabled() and z.requires_grad):
        return ops.odd_alias(z, lo, length, rowmax=rm.get("words")).contiguous()
    y = odd_length_alias(z, lo, length)
    if out is None:
        return y.contiguous()
    out.copy_(y.reshape(out.shape))
    return out


def convolve(x, h, mode="zerophase", pad_mode="min", exact=False):
    """Reference-compatible convolve(): x (R,C,L) or (R,L); h (R,Cf,N) or (R,N).  ``exact``: see reference_aliases."""
    compute_pad_len(x, h, pad_mode)
    if needs_grad(x, h):
        return diff.convolve(x, h, mode, exact=exact)
    flat = x.ndim == 2
    if flat:
        x, h = x.unsqueeze(1), h.unsqueeze(1)
    R, Cf, N = h.shape
    L = x.shape[-1]
    if N <= SHORT_FIR_TAPS and not reference_aliases(L, N, exact):
        off, Lout = {"causal": (0, L), "zerophase": (N // 2, L)}.get(mode, (0, L + N - 1))
        y = ops.fir_direct(x, h, Lout=Lout, off=off)
        return y.squeeze(1) if flat else y
    Hs = ops.fir_spectrum(h.reshape(R * Cf, N))
    y = convolve_taps(x, Hs, N, Cf, mode, exact=exact)
    return y.squeeze(1) if flat else y


class FIRConvolution(nn.Module):
    """Same constructor as the reference (convolution.py:38-65).

    ``flashfftconv=True`` (the upstream default) asks for the FlashFFTConv CUDA library; where that cannot be imported
    -- the reference's PyTorch-CPU path, and any machine with this package's HIP device -- upstream warns and uses its
    native torch.fft ``convolve`` (convolution.py:47-53).  So does this class: the flag resolves to False, the module
    computes ``convolve`` with its odd-length aliasing, and ``max_input_len`` is accepted and
    unused.  Because the flag is resolved before upstream's zero-phase check (convolution.py:55-58), zero-phase mode
    works under either value, as it does upstream on CPU."""

    def __init__(self, mode="causal", flashfftconv=True, max_input_len=2**17):
        super().__init__()
        self.mode = mode
        self.flashfftconv = resolve_flashfftconv(flashfftconv)

    def forward(self, input_signals, fir):
        return convolve(input_signals, fir, mode=self.mode, exact=self.flashfftconv)
