"""Tensor-level wrappers over the C ABI (include/grafx_amd.h).

These take torch CUDA(=HIP) tensors, hand raw device pointers and the current
stream to libgrafx_amd.so and return torch tensors.  torch is only the memory
and stream plumbing; all arithmetic happens in the HIP kernels.  CPU tensors are
rejected: there is no fallback path.
"""
import torch

from ._lib import RowMap, check, lib


def _require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "grafx_amd processors run on the MI355X only (got a CPU tensor); there is no CPU fallback. "
                "Move signals and parameters to the GPU, or use the reference/oracle on CPU."
            )
        if t.dtype != torch.float32:
            raise TypeError(f"grafx_amd kernels compute in float32, got {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def rowmap(t):
    """RowMap for a (R, C, L) tensor or a (B, n, C, L) view whose last dim is contiguous.

    Lets kernels address slices of render_grafx's signal buffer in place.
    Returns (RowMap, R, C, L).
    """
    if t.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    if t.ndim == 3:
        R, C, L = t.shape
        return RowMap(max(R, 1), 0, t.stride(0), t.stride(1)), R, C, L
    if t.ndim == 4:
        B, n, C, L = t.shape
        return RowMap(max(n, 1), t.stride(0), t.stride(1), t.stride(2)), B * n, C, L
    raise ValueError(f"expected a 3-D or 4-D signal tensor, got {t.ndim}-D")


# ----------------------------------------------------------------------------------------- FIR conv
def fir_spectrum(h, gain=None, gain_div=1):
    """Taps (RCf, N) -> opaque tile-spectrum buffer for :func:`fftconv`."""
    _require_gpu(h, gain)
    h = h.contiguous()
    RCf, N = h.shape
    Hs = torch.empty(lib().gfx_fir_spectrum_bytes(RCf, N), dtype=torch.uint8, device=h.device)
    check(lib().gfx_fir_spectrum_f32(_ptr(h), _ptr(gain), gain_div, _ptr(Hs), RCf, N, _stream()), "gfx_fir_spectrum_f32")
    return Hs


def fftconv(x, Hs, N, Cf, Lout=None, off=0, out=None):
    """y[r,c,n] = sum_k h[r,cf,k] x[r,cx,n+off-k], n < Lout (x zero outside [0,L)).

    ``x`` / ``out`` may be (R,C,L) tensors or strided (B,n,C,L) views (see :func:`rowmap`).
    """
    _require_gpu(x, out)
    xmap, R, Cin, L = rowmap(x)
    Lout = L if Lout is None else Lout
    Cout = max(Cin, Cf)
    if out is None:
        out = torch.empty((R, Cout, Lout), dtype=torch.float32, device=x.device)
    ymap, Ry, Cy, Ly = rowmap(out)
    if (Ry, Cy) != (R, Cout) or Ly < Lout:
        raise ValueError(f"output shape {tuple(out.shape)} does not match rows={R}, channels={Cout}, length>={Lout}")
    nbytes = lib().gfx_fftconv_workspace_bytes(R, Cin, L, Lout, off, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    check(
        lib().gfx_fftconv_f32(_ptr(x), xmap, _ptr(Hs), _ptr(out), ymap, R, Cin, Cf, L, Lout, off, N, _ptr(ws), nbytes, _stream()),
        "gfx_fftconv_f32",
    )
    return out
