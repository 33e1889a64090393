"""Differentiable path of the HIP processors (used only when a gradient is requested).

Inference (``torch.no_grad()`` or nothing requires grad) runs the fused HIP kernels. When autograd
needs a graph, every processor is expressed as

    parameters --(small torch ops on the GPU: activations, RBJ formulas, sampled response,
                  irfft / istft of tiny tensors)--> FIR taps h --LinearConvFn--> signal

where :class:`LinearConvFn` is the native overlap-save convolution with a **native backward**:

    y[n]      = sum_k h[k] x[n + off - k]                                  gfx_fftconv_f32
    grad_x[m] = sum_k h[k] g[m - off + k]  = linconv(g, flip(h))[m + N-1-off]   (same kernel)
    grad_h[k] = sum_n g[n] x[n + off - k]  = linconv(g, flip(x))[k + L-1-off]   (same kernel,
                                              the reversed signal is the partitioned "filter")

so the heavy O(R*L) work of both passes stays in the HIP kernels; only the light front-end math
(R x K x N/2 complex, R x 193 x 313, elementwise gain curves) is left to torch's autograd.
The formulas mirror the reference (file:line cited per function, relative to
/root/reference/src/grafx/processors).
"""
import math

import threading

import torch
import torch.nn.functional as F

from . import ops


# Set by the stage-wise backward of render_grafx while it re-traces a stage only to obtain its tape: the stage's
# output VALUES are then never read (only its gradient function runs), so autograd nodes whose output is the
# processor's output up to linear operations may skip their forward kernels and hand back uninitialised storage.
# Thread-local (the re-trace runs in the thread that opened the scope), and opt-in per processor TYPE: the render only
# opens the scope for the exact classes of TAPE_SAFE_TYPES -- a user subclass that post-processes super().forward()
# non-linearly must get real values.
_STATE = threading.local()


def tape_only_active():
    return getattr(_STATE, "tape_only", False)


def tape_placeholder(shape, device):
    """What a node hands back as its output while `tape_only` is active: a tensor of the right shape whose values nobody
    reads -- ONE element expanded with zero strides, so that it occupies no memory (the full-size outputs it stands in for
    are 8 GiB per stage at the headline batch: two of them were a fifth of the training step's peak)."""
    return torch.empty((1,), dtype=torch.float32, device=device).expand(shape)


class tape_only:
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        self.prev = tape_only_active()
        _STATE.tape_only = self.prev or self.enabled

    def __exit__(self, *exc):
        _STATE.tape_only = self.prev
        return False


# Gradient sinks (registered by the same stage-wise backward): data_ptr of a stage's input view -> destination view.  An
# autograd node whose input IS that view writes its input gradient straight into the destination -- a slice of the
# render's gradient buffer -- instead of a fresh tensor that would then be copied there.  Keyed by address under a lock,
# so two renders running their backward passes on different threads (different buffers) cannot see each other's sinks;
# every sink counts how often it was written, and the render checks that it was written exactly once.
_SINKS = {}
_SINKS_LOCK = threading.Lock()


class grad_sink:
    def __init__(self, x, dest):
        self.key, self.entry = x.data_ptr(), [dest, 0]

    def __enter__(self):
        with _SINKS_LOCK:
            _SINKS[self.key] = self.entry
        return self

    def __exit__(self, *exc):
        with _SINKS_LOCK:
            _SINKS.pop(self.key, None)
        return False

    @property
    def writes(self):
        return self.entry[1]


def _sink_for(x):
    with _SINKS_LOCK:
        e = _SINKS.get(x.data_ptr())
        if e is not None and tuple(x.shape) == tuple(e[0].shape):
            e[1] += 1
            return e[0]
    return None


# Gradient sources (the mirror image of the sinks; same registration by the stage-wise backward): data_ptr of a stage's
# input view -> the gradient of the stage's OUTPUT, held in a layout torch cannot express as a tensor of the output's shape
# -- a (B k, m, C, L) view with a zero stride over m: k distinct gradient rows per graph, each shared by m consecutive
# output rows (the adjoint of a routing sum whose sources feed the same destinations in blocks, e.g. the eight channel
# strips of a console bus).  The node whose input IS that view reads its output gradient from the source through the row
# map (the kernels address rows as (r / inner) * stride_outer + (r % inner) * stride_inner: stride_inner = 0) and ignores
# the placeholder the engine hands it; the 8.6 GB of expanded rows at the headline batch are never written or read.
_SOURCES = {}


class grad_source:
    def __init__(self, x, source):
        self.key, self.entry = x.data_ptr(), [source, 0]

    def __enter__(self):
        with _SINKS_LOCK:
            _SOURCES[self.key] = self.entry
        return self

    def __exit__(self, *exc):
        with _SINKS_LOCK:
            _SOURCES.pop(self.key, None)
        return False

    @property
    def reads(self):
        return self.entry[1]


def _source_for(x, rows):
    with _SINKS_LOCK:
        e = _SOURCES.get(x.data_ptr())
        if e is not None and e[0].shape[0] * e[0].shape[1] == rows and tuple(e[0].shape[2:]) == tuple(x.shape[-2:]):
            e[1] += 1
            return e[0]
    return None


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class LinearConvFn(torch.autograd.Function):
    """y[r,c,n] = sum_k h[r % Rh,cf,k] x[r,cx,n+off-k], n in [0,Lout); channels broadcast 1<->2.

    ``h`` may hold fewer rows than ``x`` (Rh divides R; rows are batch-major, so Rh = nodes shares one filter per
    node across the batch): the spectra are built once per filter, and grad_h is the per-row gradient summed over
    the batch."""

    @staticmethod
    def forward(ctx, x, h, Lout, off, final=False):
        # x: (R,C,L), or a strided (B,n,C,L) view of the signal buffer, read in place through its row map; the
        # output then comes back as (B,n,C,Lout) so that a strided 4-D gradient can be fed in without a copy
        four = x.ndim == 4
        if x.stride(-1) != 1 or not (four or x.is_contiguous()):
            x = x.contiguous()
        h = h.contiguous()
        Rh, Cf, N = h.shape
        rows = x.shape[0] * x.shape[1] if four else x.shape[0]
        if rows % Rh != 0:
            raise ValueError(f"{rows} signal rows cannot share {Rh} filters")
        ctx.save_for_backward(x, h)
        ctx.off = off
        if final and tape_only_active():  # see tape_only: nobody will read these values
            y = tape_placeholder((rows, max(x.shape[-2], Cf), Lout), x.device)
        else:
            y = ops.fftconv(x, ops.fir_spectrum(h.reshape(Rh * Cf, N)), N, Cf, Lout=Lout, off=off, h_rows=Rh)
        return y.view(x.shape[0], x.shape[1], y.shape[1], Lout) if four else y

    @staticmethod
    def backward(ctx, g):
        x, h = ctx.saved_tensors
        off = ctx.off
        four = x.ndim == 4
        if g.stride(-1) != 1 or not (g.ndim == 4 or g.is_contiguous()):
            g = g.contiguous()
        Cin, L = x.shape[-2], x.shape[-1]
        R = x.shape[0] * x.shape[1] if four else x.shape[0]
        Rh, Cf, N = h.shape
        gx = gh = None
        if ctx.needs_input_grad[0]:
            hr = h.flip(-1).contiguous()
            sink = _sink_for(x) if max(Cin, Cf) == Cin else None
            gx = ops.fftconv(g, ops.fir_spectrum(hr.reshape(Rh * Cf, N)), N, Cf, Lout=L, off=N - 1 - off, h_rows=Rh,
                             out=sink)
            if sink is None:
                if gx.shape[1] != Cin:  # x was broadcast over the output channels
                    gx = gx.sum(1, keepdim=True)
                gx = gx.view(x.shape)
        if ctx.needs_input_grad[1]:
            if N <= 8193 and abs(off) <= 8192:  # short filter: tile-wise correlation, x and g read once
                gh = ops.fir_grad(x, g, N, off)
            else:
                P = ops.part_len_for(L, N)  # long signal as the filter, N outputs: fewer, longer partitions
                # the reversed signal is the filter of this correlation; its spectra are taken from x in place
                gh = ops.fftconv(g, ops.fir_spectrum_reversed(x, part_len=P), L, Cin, Lout=N, off=L - 1 - off,
                                 part_len=P)
            if gh.shape[1] != Cf:  # one filter shared by both channels
                gh = gh.sum(1, keepdim=True)
            if Rh != R:  # one filter shared by the batch
                gh = gh.view(R // Rh, Rh, Cf, N).sum(0)
        return gx, gh, None, None, None


class OddAliasFn(torch.autograd.Function):
    """y = irfft_{P-1}(rfft_P(z))[..., lo : lo + length] (reference core/convolution.py:123-126) on the chirp-z kernels,
    its gradient on the transposed pair (gfx_odd_alias_adjoint_f32): no FFT library, no float64."""

    @staticmethod
    def forward(ctx, z, lo, length, precise):
        ctx.P, ctx.lo, ctx.precise = z.shape[-1], lo, precise
        return ops.odd_alias(z, lo, length, precise=precise)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        return ops.odd_alias_adjoint(gy, ctx.P, ctx.lo, precise=ctx.precise), None, None, None


def odd_alias(z, lo=0, length=None, precise=False):
    """Differentiable odd-length aliasing (``precise``: double-precision transforms, see
    processors.core.convolution.odd_length_alias)."""
    Q = z.shape[-1] - 1
    length = Q - lo if length is None else length
    if not ops.odd_alias_supported(z.shape[-1]):
        raise NotImplementedError(f"odd-length aliasing: P = {z.shape[-1]} is beyond the kernels' range (P <= 11,184,811)")
    return OddAliasFn.apply(z, lo, length, bool(precise))


def convolve(x, h, mode="causal", exact=False, final=False, precise=False):
    """Differentiable twin of processors.core.convolution.convolve (reference core/convolution.py:119-134),
    including the odd-P aliasing (OddAliasFn).  ``final``: the result is the calling processor's output up to linear
    operations (see tape_only)."""
    from .processors.core.convolution import reference_aliases

    flat = x.ndim == 2
    if flat:
        x, h = x.unsqueeze(1), h.unsqueeze(1)
    L, N = x.shape[-1], h.shape[-1]
    if not reference_aliases(L, N, exact):
        if mode == "causal":
            y = LinearConvFn.apply(x, h, L, 0, final)
        elif mode == "zerophase":
            y = LinearConvFn.apply(x, h, L, N // 2, final)
        else:
            y = LinearConvFn.apply(x, h, L + N - 1, 0, final)
    else:
        z = LinearConvFn.apply(x, h, L + N - 1, 0, final)
        if mode == "causal":
            y = odd_alias(z, 0, L, precise)
        elif mode == "zerophase":
            y = odd_alias(z, N // 2, L, precise)
        else:
            y = odd_alias(z, precise=precise)
    return y.squeeze(1) if flat else y


# ----------------------------------------------------------------------------------------- front-ends
_CHIRPS = {}


def _irfft_odd(resp, N):
    """irfft(resp, n=N) for odd N as a chirp-z transform on power-of-two FFTs (differentiable torch ops).

    The FFT library runs odd lengths through its own Bluestein plans, which cannot be recorded into a HIP graph;
    spelled out here, the training step stays capturable.  With w_j = exp(i pi j^2 / N) (angles reduced in integer
    arithmetic, evaluated in float64):  x[n] = (2/N) Re( w_n * sum_k (b_k X[k] w_k) conj(w_{n-k}) ),  b_0 = 1/2."""
    M = N // 2
    P2 = 1 << (N + M - 1).bit_length()
    key = (N, resp.device.type, resp.device.index)
    if key not in _CHIRPS:
        j = torch.arange(-M, N, dtype=torch.int64)
        w = torch.polar(torch.ones(j.numel(), dtype=torch.float64), math.pi * ((j * j) % (2 * N)).double() / N)
        v = torch.zeros(P2, dtype=torch.complex128)
        v[:N] = w[M:].conj()
        v[P2 - M:] = w[:M].conj()
        scale = torch.full((M + 1,), 1.0, dtype=torch.float64)
        scale[0] = 0.5
        _CHIRPS[key] = ((w[M : 2 * M + 1] * scale).to(torch.complex64).to(resp.device),       # b_k w_k
                        torch.fft.fft(v).to(torch.complex64).to(resp.device),                  # FFT of conj(w_{j})
                        (w[M:] * (2.0 / N)).to(torch.complex64).to(resp.device))               # (2/N) w_n
    wk, V, wn = _CHIRPS[key]
    S = torch.fft.ifft(torch.fft.fft(resp * wk, n=P2, dim=-1) * V, dim=-1)[..., :N]
    return (S * wn).real


def fsm_fir(Bs, As, fir_len):
    """core/iir.py:147-150, 263-276: sampled cascade response -> irfft(n=N)."""
    d = torch.arange(Bs.shape[-1], device=Bs.device)
    k = torch.arange(fir_len // 2 + 1, device=Bs.device)
    phase = (d[:, None] * k[None, :]).to(Bs.dtype) / fir_len * 2 * math.pi
    delays = torch.exp(-1j * phase)
    sections = (Bs.unsqueeze(-1) * delays).sum(-2) / (As.unsqueeze(-1) * delays).sum(-2)
    resp = sections[..., 0, :]
    for i in range(1, sections.shape[-2]):  # not .prod(): its backward asks the host whether any factor is zero
        resp = resp * sections[..., i, :]
    if resp.is_cuda and resp.dtype == torch.complex64 and fir_len <= ops.IRDFT_MAX_N:
        return irfft_small(resp, fir_len)       # any length up to 8192: direct-sum kernels both ways, no FFT library
    if resp.is_cuda:
        ops.fft_library_reached(f"frequency-sampled taps of {fir_len} points (differentiable design)")
    if fir_len % 2 == 1 and resp.is_cuda:
        return _irfft_odd(resp, fir_len)
    return torch.fft.irfft(resp, dim=-1, n=fir_len)


class IrdftFn(torch.autograd.Function):
    """irfft(X, n) along the last axis on the direct-sum kernels (n <= 8192): forward gfx_irdft_f32, backward its
    adjoint, (c_k / n) rfft(g) with c = (1, 2, 2, ..., [1 at n/2 for even n]) on gfx_rdft_f32 -- what torch.fft.irfft
    computes and what autograd derives from it, without the FFT library."""

    @staticmethod
    def forward(ctx, X, n):
        ctx.n = n
        return ops.irdft(X.contiguous(), n)

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        G = ops.rdft(g.contiguous(), n) * (2.0 / n)
        G[..., 0] = G[..., 0] * 0.5
        if n % 2 == 0:
            G[..., -1] = G[..., -1] * 0.5
        return G, None


def irfft_small(X, n):
    """torch.fft.irfft(X, n=n, dim=-1) for complex64 X on the GPU and n <= 8192, off the FFT library (differentiable)."""
    if X.is_cuda and X.dtype == torch.complex64 and n <= ops.IRDFT_MAX_N:
        return IrdftFn.apply(X, n)
    if X.is_cuda:
        ops.fft_library_reached(f"inverse real DFT of {n} points (differentiable design)")
    return torch.fft.irfft(X, n=n, dim=-1)


class PeqCoeffsFn(torch.autograd.Function):
    """eq.py:291-314 + filter.py:593-754 as one native kernel each way (gfx_peq_coeffs_f32 / _bwd_f32) instead of the
    ~40 + ~120 elementwise kernels torch autograd makes of :func:`peq_coefficients`."""

    @staticmethod
    def forward(ctx, w0, q_inv, log_gain, use_shelving_filters):
        ctx.save_for_backward(w0, q_inv, log_gain)
        ctx.shelving = use_shelving_filters
        return ops.peq_coeffs(w0, q_inv, log_gain, use_shelving_filters)

    @staticmethod
    def backward(ctx, gBs, gAs):
        w0, q_inv, log_gain = ctx.saved_tensors
        gBs = torch.zeros_like(w0).unsqueeze(-1).expand(*w0.shape, 3) if gBs is None else gBs
        gAs = torch.zeros_like(w0).unsqueeze(-1).expand(*w0.shape, 3) if gAs is None else gAs
        return (*ops.peq_coeffs_bwd(w0, q_inv, log_gain, gBs, gAs, ctx.shelving), None)


_DELAYS = {}


def _fsm_delays(N, device):
    """exp(-j 2 pi d k / N), d = 0..2, k = 0..N//2 with the float32 phase upstream uses (core/iir.py:263-276)."""
    key = (N, device.type, device.index)
    if key not in _DELAYS:
        d = torch.arange(3, device=device)
        k = torch.arange(N // 2 + 1, device=device)
        _DELAYS[key] = torch.exp(-1j * ((d[:, None] * k[None, :]).to(torch.float32) / N * 2 * math.pi))
    return _DELAYS[key]


FSM_BWD_NATIVE = True     # FsmFirFn's backward on gfx_iir_fsm_bwd_f32 (False: the batched torch ops of rounds 4-5)


class FsmFirFn(torch.autograd.Function):
    """(R,Cf,K,3) biquad coefficients -> (R,Cf,N) frequency-sampled taps (core/iir.py:147-150): forward is the native
    response + Bluestein kernel; backward is written out (a dozen batched complex ops instead of autograd's ~80):

        h = irfft(resp),  resp = prod_i num_i / den_i,  num_i = sum_d B[i,d] D_d
        dL/dB[i,d] =  Re sum_k conj(G_k) resp_k D_d[k] / num_i[k],   G = (c_k / N) rfft(dL/dh),  c = (1, 2, 2, ...)
        dL/dA[i,d] = -Re sum_k conj(G_k) resp_k D_d[k] / den_i[k]
    """

    @staticmethod
    def forward(ctx, Bs, As, N, plan):
        ctx.save_for_backward(Bs, As)
        ctx.N = N
        R, Cf = Bs.shape[0], Bs.shape[1]
        return ops.iir_fsm_fir(Bs, As, N, plan).view(R, Cf, N)

    @staticmethod
    def backward(ctx, gh):
        Bs, As = ctx.saved_tensors
        N = ctx.N
        if (FSM_BWD_NATIVE and gh.is_cuda and N <= ops.IRDFT_MAX_N and Bs.dtype == torch.float32 and Bs.shape[-1] == 3):
            # round 6: the bins' sums in ONE native launch, double precision inside (gfx_iir_fsm_bwd_f32), after the native
            # real DFT of the incoming gradient -- ~40 complex128 torch kernels per equaliser stage and step before
            G = ops.rdft(gh.contiguous(), N)
            gB, gA = ops.iir_fsm_bwd(Bs, As, G, _fsm_delays(N, Bs.device), N, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
            return gB, gA, None, None
        return FsmFirFn.backward_torch(Bs, As, gh, N, ctx.needs_input_grad[0], ctx.needs_input_grad[1])

    @staticmethod
    def backward_torch(Bs, As, gh, N, want_B=True, want_A=True):
        """The same gradient as batched torch ops (any length; the reference the native kernel is tested against)."""
        # Parameter-side math ((R, Cf, K, F) values, F <= 2049) in DOUBLE precision since round 5: the sums over the bins
        # cancel to a few 1e-5 of their terms, and in complex64 the w0 / q_inv gradients behind them carried 0.9 .. 1.6e-5
        # of rounding -- as much as the reference's own float32 autograd (0.7 .. 1.5e-5), so "at least as close to
        # float64 as the reference" was a coin toss (tests/test_gpu_autograd.py::test_peq_parameter_gradients_vs_reference).
        # Same sample points as the forward pass (the reference's float32 phase), widened.
        out_dtype = Bs.dtype
        Bs, As, gh = Bs.double(), As.double(), gh
        D = _fsm_delays(N, Bs.device).to(torch.complex128)   # (3, F)
        # three-term sums written out: the library's complex GEMM takes 0.16 ms for these 3-wide contractions
        num = Bs[..., 0:1] * D[0] + Bs[..., 1:2] * D[1] + Bs[..., 2:3] * D[2]   # (R,Cf,K,F)
        den = As[..., 0:1] * D[0] + As[..., 1:2] * D[1] + As[..., 2:3] * D[2]
        sections = num / den
        resp = sections[..., 0, :]
        for i in range(1, sections.shape[-2]):
            resp = resp * sections[..., i, :]
        if N > ops.IRDFT_MAX_N and gh.is_cuda:
            ops.fft_library_reached(f"gradient of frequency-sampled taps of {N} points")
        G = (ops.rdft(gh, N) if N <= ops.IRDFT_MAX_N else torch.fft.rfft(gh, n=N, dim=-1)).to(torch.complex128) * (2.0 / N)
        G[..., 0] = G[..., 0] * 0.5
        if N % 2 == 0:
            G[..., -1] = G[..., -1] * 0.5
        T = (G.conj() * resp).unsqueeze(-2)                # (R,Cf,1,F)

        def contract(Q):  # Re sum_k Q[..., k] D_d[k], d = 0..2  ->  (..., 3)
            return torch.stack([(Q * D[d]).real.sum(-1) for d in range(3)], -1)

        gB = contract(T / num).to(out_dtype) if want_B else None
        gA = (-contract(T / den)).to(out_dtype) if want_A else None
        return gB, gA, None, None


def peq_coefficients(w0, q_inv, log_gain, use_shelving_filters=True):
    """eq.py:291-314 + filter.py:593-604, 645-656, 687-705, 736-754."""
    w = math.pi * torch.sigmoid(w0)
    A = torch.exp(log_gain)
    cw = torch.cos(w)
    alpha = torch.sin(w) * torch.exp(q_inv) * 0.5

    def peaking(cw, al, A):
        return (torch.stack([1 + al * A, -2 * cw, 1 - al * A], -1), torch.stack([1 + al / A, -2 * cw, 1 - al / A], -1))

    def shelf(cw, al, A, sg):
        ap1, am1 = A + 1, A - 1
        s = 2 * A.sqrt() * al
        b = torch.stack([A * (ap1 - sg * am1 * cw + s), sg * 2 * A * (am1 - sg * ap1 * cw), A * (ap1 - sg * am1 * cw - s)], -1)
        a = torch.stack([ap1 + sg * am1 * cw + s, -sg * 2 * (am1 + sg * ap1 * cw), ap1 + sg * am1 * cw - s], -1)
        return b, a

    if not use_shelving_filters:
        return peaking(cw, alpha, A)
    K = w0.shape[-1]
    parts = [shelf(cw[..., :1], alpha[..., :1], A[..., :1], 1.0),
             peaking(cw[..., 1 : K - 1], alpha[..., 1 : K - 1], A[..., 1 : K - 1]),
             shelf(cw[..., K - 1 :], alpha[..., K - 1 :], A[..., K - 1 :], -1.0)]
    return torch.cat([p[0] for p in parts], -2), torch.cat([p[1] for p in parts], -2)


def biquad_coefficients(Bs, A1_pre, A2_pre, A0=None):
    """filter.py:144-153."""
    a1 = 2 * torch.tanh(A1_pre)
    a2 = ((2 - a1.abs()) * torch.tanh(A2_pre) + a1.abs()) / 2
    As = torch.stack([torch.ones_like(a1), a1, a2], -1)
    if A0 is not None:
        As = As * A0.unsqueeze(-1)
    Bs = torch.cat([Bs[..., :1] + 1, Bs[..., 1:]], -1)
    return Bs.unsqueeze(1), As.unsqueeze(1)


def one_pole_fir(z_alpha, iir_len):
    """core/envelope.py:51-60."""
    alpha = torch.sigmoid(z_alpha).clamp(max=1 - 1e-5)
    n = torch.arange(iir_len, device=z_alpha.device)[None, :]
    return (1 - alpha) * torch.exp(n * torch.log(alpha))


class TruncatedOnePoleFn(torch.autograd.Function):
    """y = relu(u * h), h[k] = (1-a) a^k, k < N, a = min(sigmoid(z), 1-1e-5)  (core/envelope.py:34-60), as the
    native prefix scan in both directions instead of a 16384-tap FFT convolution and its two adjoint convolutions.

    With U[n] = sum_{k<=n} a^k u[n-k] (the untruncated scan) the filter output is (1-a)(U[n] - a^N U[n-N]), so
      grad_u[m] = sum_{k<N} h[k] g[m+k]                      -- the same scan run backwards in time
      d/da      = -(U[n] - a^N U[n-N]) + (1-a)(D[n] - N a^(N-1) U[n-N] - a^N D[n-N]),
                  D = dU/da = sum_k k a^(k-1) u[n-k],  D[n] = a D[n-1] + U[n-1]   -- one more scan
    (g already masked by the relu)."""

    @staticmethod
    def forward(ctx, u, z_alpha, N):
        u = u.contiguous()
        y = ops.onepole(u, z_alpha, N, relu=True)
        ctx.save_for_backward(u, z_alpha, y)
        ctx.N = N
        return y

    @staticmethod
    def backward(ctx, g):
        u, z_alpha, y = ctx.saved_tensors
        gu, gz = one_pole_backward(u, z_alpha, y, ctx.N, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gu, gz, None


def one_pole_backward(u, z_alpha, y, N, g, need_u=True, need_z=True):
    """Adjoint of y = relu(truncated one-pole(u)) given g = dL/dy (see TruncatedOnePoleFn)."""
    R, L = u.shape
    g = (g * (y > 0)).contiguous()
    gu = gz = None
    if need_u:
        gu = ops.onepole(g.flip(-1), z_alpha, N, relu=False).flip(-1)
    if need_z:
        gz = pole_gradient(g, ops.onepole(u, z_alpha, L + 1, relu=False), z_alpha, N)
    return gu, gz


def pole_gradient(g, U1, z_alpha, N):
    """dL/dz_alpha of the truncated one-pole given the (relu-masked) output gradient g and U1 = (1-a) * (the
    un-truncated scan of the input).  With a^N = aN:
        da = sum_n g[n] ( -U[n] + (aN - (1-a) N a^(N-1)) U[n-N] + (1-a) (D[n] - aN D[n-N]) ),   D[n] = S[n-1],
    S = scan of U.  The scan kernel returns S2 = (1-a)^2 S from U1; the powers of 1/(1-a) go into the coefficients
    and the one-sample shift of D onto g."""
    R, L = g.shape
    sig = torch.sigmoid(z_alpha.reshape(R, 1))
    a = sig.clamp(max=1 - 1e-5)
    one_m_a = 1 - a
    S2 = ops.onepole(U1, z_alpha, L + 1, relu=False)  # iir_len > L: no truncation term
    aN = torch.pow(a.double(), N).float()
    aN1 = torch.pow(a.double(), N - 1).float()
    inv, zero = 1 / one_m_a, torch.zeros_like(a)
    da = ops.onepole_dz(g, U1, S2, torch.cat([-inv, inv, (aN - one_m_a * N * aN1) * inv, -aN * inv], 1), N)
    return (da.reshape(R, 1) * sig * (1 - sig) * (sig < 1 - 1e-5)).reshape(z_alpha.shape)


class DynamicsFn(torch.autograd.Function):
    """Compressor / NoiseGate without gain smoother (dynamics.py:390-405, 625-640) as one autograd node.

    Forward is the fused inference kernel.  Backward recomputes the (smoothed) energy with the native energy and
    scan kernels, takes the gain computer's derivatives in one native pass (gfx_dyn_gain_bwd_f32), pushes the
    energy gradient back through the one-pole smoother with the scan run backwards in time, and assembles the
    input gradient in one more pass (gfx_dyn_dx_f32) -- instead of ~40 full-size elementwise torch kernels and
    three 16384-tap FFT convolutions."""

    @staticmethod
    def forward(ctx, x, log_threshold, log_ratio, log_knee, z_alpha, smoother, iir_len, knee, gate, u1=None):
        # u1: the smoother's un-truncated scan of this very input, kept by an earlier forward pass (the stage-wise
        # backward of render_grafx hands over what the inference render stored); None: produced / recomputed here
        four = x.ndim == 4  # a strided (B,n,C,L) view of the signal buffer is read in place
        if x.stride(-1) != 1 or not (four or x.is_contiguous()):
            x = x.contiguous()
        rows = x.shape[0] * x.shape[1] if four else x.shape[0]
        if tape_only_active():  # the output of this node is the processor's output: its values are not read (see tape_only)
            y = tape_placeholder((rows, x.shape[-2], x.shape[-1]), x.device)
        else:
            if smoother and u1 is None:
                u1 = torch.empty((rows, x.shape[-1]), dtype=torch.float32, device=x.device)
                y = ops.dynamics_fused(x, log_threshold, log_ratio, log_knee if knee != "hard" else None, z_alpha,
                                       smoother=1, iir_len=iir_len, knee=knee, gate=gate, u1_out=u1)
            else:
                y = ops.dynamics_fused(x, log_threshold, log_ratio, log_knee if knee != "hard" else None, z_alpha,
                                       smoother=int(smoother), iir_len=iir_len, knee=knee, gate=gate)
        ctx.save_for_backward(x, log_threshold, log_ratio, log_knee, z_alpha)
        ctx.cfg = (smoother, iir_len, knee, gate)
        ctx.u1 = u1 if smoother else None
        return y.view(x.shape) if four else y

    @staticmethod
    def backward(ctx, gy):
        x, log_threshold, log_ratio, log_knee, z_alpha = ctx.saved_tensors
        smoother, iir_len, knee, gate = ctx.cfg
        src = _source_for(x, x.shape[0] * x.shape[1] if x.ndim == 4 else x.shape[0])
        if src is not None:     # the real gradient, in a layout only the row map expresses (see grad_source)
            gy = src
        if gy.stride(-1) != 1 or not (gy.ndim == 4 or gy.is_contiguous()):
            gy = gy.contiguous()
        lk = log_knee if knee != "hard" else None
        if smoother:  # two fused passes over the rows (forward, then backward in time) + the pole-gradient reduction
            sink = _sink_for(x)
            gx, gp, da = ops.dynamics_bwd(x, gy, log_threshold, log_ratio, lk, z_alpha, iir_len, knee, gate,
                                          out=sink, pole=ctx.needs_input_grad[4], u1=ctx.u1)
            ctx.u1 = None
            gz = None
            if da is not None:  # chain rule through a = min(sigmoid(z), 1 - 1e-5)
                sig = torch.sigmoid(z_alpha.reshape(-1))
                gz = (da * sig * (1 - sig) * (sig < 1 - 1e-5)).reshape(z_alpha.shape)
            gx = gx.view(x.shape)
        else:
            e = ops.energy(x)
            gain, denv, gp = ops.dyn_gain_bwd(x, gy, e, log_threshold, log_ratio, lk, knee, gate)
            gx = ops.dyn_dx(x, gy, gain, denv).view(x.shape) if ctx.needs_input_grad[0] else None
            gz = None
        like = lambda t, col: None if t is None else gp[:, col].reshape(t.shape)  # noqa: E731
        return (gx, like(log_threshold, 0), like(log_ratio, 1), like(log_knee if knee != "hard" else None, 2), gz,
                None, None, None, None, None)


def truncated_one_pole(u, z_alpha, iir_len, exact=False):
    """core/envelope.py:34-49."""
    from .processors.core.convolution import reference_aliases

    if u.ndim == 2 and not reference_aliases(u.shape[-1], iir_len, exact):
        return TruncatedOnePoleFn.apply(u, z_alpha, iir_len)
    return torch.relu(convolve(u, one_pole_fir(z_alpha, iir_len), "causal", exact=exact, precise=True))


class BallisticsFn(torch.autograd.Function):
    """Attack/release smoother (core/envelope.py:84-101 -> torchcomp.compressor_core, semantics as recalled):
    native forward recursion and native adjoint recursion (run backwards in time over the saved output)."""

    @staticmethod
    def forward(ctx, x, z_alpha):
        x, z_alpha = x.contiguous(), z_alpha.contiguous()
        y = ops.ballistics(x, z_alpha)
        ctx.save_for_backward(x, z_alpha, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, z_alpha, y = ctx.saved_tensors
        gx, gz = ops.ballistics_bwd(x, y, g, z_alpha)
        return gx, gz.reshape(z_alpha.shape)


def log_gain(G, T, log_ratio, log_knee, knee, gate):
    """dynamics.py:444-489 (compressor), 676-721 (gate)."""
    R = 1 + torch.exp(log_ratio)
    if not gate:
        if knee == "hard":
            return torch.minimum(G, T + (G - T) / R) - G
        if knee == "quadratic":
            W = torch.exp(log_knee) / 2
            below, above = G < (T - W), G > (T + W)
            mid = ~below & ~above
            out = G * below + (T + (G - T) / R) * above + (G + (1 / R - 1) * (G - T + W).square() / (4 * W)) * mid
            return out - G
        k = torch.exp(log_knee)
        return (1 / R - 1) * F.softplus(k * (G - T)) / k
    if knee == "hard":
        return torch.minimum(G, R * (G - T) + T) - G
    if knee == "quadratic":
        W = torch.exp(log_knee) / 2
        below, above = G < (T - W), G > (T + W)
        mid = ~below & ~above
        out = (R * (G - T) + T) * below + G * above + (G + (1 - R) * (G - T - W).square() / (4 * W)) * mid
        return out - G
    k = torch.exp(log_knee)
    return -torch.exp(log_ratio) * F.softplus(k * (T - G)) / k


# ------------------------------------------------------------------ exact recursive biquad cascade (lfilter / ssm backends)
class BiquadCascadeFn(torch.autograd.Function):
    """y = H_K ... H_1 x with H_i = B_i(z) / A_i(z), the exact recursion of gfx_biquad_cascade_f32 (reference
    core/iir.py:154-183: one torchaudio.lfilter call per section, differentiable upstream).

    Backward, per section from the last to the first (g_K = dL/dy, u_0 = x, u_i = H_i u_{i-1} saved by the forward):
        v_i     = (1 / A_i)^T g_i                    the all-pole part run backwards in time (same native kernel on the
                                                     time-reversed gradient)
        dB_i[d] =  sum_n v_i[n] u_{i-1}[n - d]       d = 0, 1, 2
        dA_i[d] = -sum_n v_i[n] u_i[n - d]           d = 0, 1, 2   (dy/da_d = -(z^-d / A) y)
        g_{i-1} = B_i^T v_i                          g_{i-1}[n] = sum_d b_d v_i[n + d]
    and dL/dx = g_0.  Channel broadcasts (1 <-> C) reduce the corresponding gradient over the channel axis."""

    @staticmethod
    def forward(ctx, x, Bs, As):
        K = Bs.shape[2]
        us = [x.contiguous()]
        for i in range(K):
            us.append(ops.biquad_cascade(us[-1], Bs[:, :, i : i + 1].contiguous(), As[:, :, i : i + 1].contiguous()))
        ctx.save_for_backward(Bs, As, *us)
        return us[-1]

    @staticmethod
    def backward(ctx, gy):
        Bs, As, *us = ctx.saved_tensors
        R, Cf, K, _ = Bs.shape
        L = gy.shape[-1]
        g = gy.contiguous()
        Cout = g.shape[1]
        gB, gA = torch.zeros_like(Bs), torch.zeros_like(As)
        unit = torch.zeros((R, Cf, 1, 3), dtype=Bs.dtype, device=Bs.device)
        unit[..., 0] = 1.0

        def to_filter_channels(t):     # (R, Cout) -> (R, Cf)
            return t.sum(1, keepdim=True) if (Cf == 1 and Cout > 1) else t

        for i in reversed(range(K)):
            Ai = As[:, :, i : i + 1].contiguous()
            v = ops.biquad_cascade(g.flip(-1).contiguous(), unit, Ai).flip(-1)   # (R, Cout, L)
            u_in, u_out = us[i], us[i + 1]
            for d in range(3):
                vd = v[..., d:]
                gB[:, :, i, d] = to_filter_channels((vd * u_in[..., : L - d]).sum(-1).expand(R, Cout))
                gA[:, :, i, d] = -to_filter_channels((vd * u_out[..., : L - d]).sum(-1))
            b = Bs[:, :, i]                                                        # (R, Cf, 3), broadcasts over channels
            gp = b[..., 0:1] * v
            gp[..., :-1] += b[..., 1:2] * v[..., 1:]
            gp[..., :-2] += b[..., 2:3] * v[..., 2:]
            g = gp
        x = us[0]
        gx = g.sum(1, keepdim=True) if x.shape[1] == 1 and Cout > 1 else g
        return (gx if ctx.needs_input_grad[0] else None, gB if ctx.needs_input_grad[1] else None,
                gA if ctx.needs_input_grad[2] else None)
