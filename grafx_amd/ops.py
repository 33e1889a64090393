"""Tensor-level wrappers over the C ABI (include/grafx_amd.h).

These take torch CUDA(=HIP) tensors, hand raw device pointers and the current
stream to libgrafx_amd.so and return torch tensors.  torch is only the memory
and stream plumbing; all arithmetic happens in the HIP kernels.  CPU tensors are
rejected: there is no fallback path.
"""
import contextvars
import ctypes
import os

import torch

from ._lib import RowMap, check, lib


# Per-launch timing hook: `with ops.profiling() as prof:` collects (start_event, end_event, algorithmic_bytes) per kernel
# launch issued from the current context (bench.py's live roofline measurement).  A ContextVar, not a module global:
# another thread rendering at the same time neither pays for the events nor pollutes the record.
_PROFILE = contextvars.ContextVar("grafx_amd_profile", default=None)


class profiling:
    def __enter__(self):
        self.record = {}
        self.token = _PROFILE.set(self.record)
        return self.record

    def __exit__(self, *exc):
        _PROFILE.reset(self.token)
        return False


class _timed:
    def __init__(self, name, nbytes):
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        self.rec = _PROFILE.get()
        if self.rec is not None:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.b.record()
            self.rec.setdefault(self.name, []).append((self.a, self.b, self.nbytes))


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


class _Pin:
    """Pointer factory for ONE launch: ``pin(t)`` returns the device address of a contiguous version of ``t`` and
    keeps that (possibly temporary) tensor referenced until the wrapper returns, i.e. until the launch is enqueued.
    Taking ``data_ptr()`` of an unnamed ``.contiguous()`` result frees the temporary before the launch; the next
    same-size temporary then reuses the block and two arguments silently alias (round-1 advisor finding)."""

    def __init__(self):
        self.keep = []

    def __call__(self, t):
        if t is None:
            return 0
        t = t.contiguous()
        self.keep.append(t)
        return t.data_ptr()


def _on_device(fn):
    """Run a wrapper with the tensors' device current: the stream handed to the library (``_stream()``) and the
    library's per-device tables (``hipGetDevice`` in csrc/common.hip) both follow the *current* device, so tensors on
    cuda:1 under a current device cuda:0 would be launched on the wrong device.  All tensor arguments must share
    one device."""
    import functools

    @functools.wraps(fn)
    def run(*args, **kwargs):
        dev = None
        for a in (*args, *kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise ValueError(f"{fn.__name__}: tensors on different devices ({dev} and {a.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)

    return run


def _expect(t, shape, what):
    """The kernels trust their sizes: every secondary tensor's shape is checked here, before its pointer is passed."""
    if t is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}")


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
def part_len_for(N, Lout):
    """Partition length the library prefers for an N-tap filter producing Lout samples per row (0 = default)."""
    return lib().gfx_fftconv_part_len(N, Lout)


@_on_device
def fir_spectrum(h, gain=None, gain_div=1, part_len=0):
    """Taps (RCf, N) -> opaque tile-spectrum buffer for :func:`fftconv` (same ``part_len`` there)."""
    _require_gpu(h, gain)
    h = h.contiguous()
    RCf, N = h.shape
    Hs = torch.empty(lib().gfx_fir_spectrum_bytes_ex(RCf, N, part_len), dtype=torch.uint8, device=h.device)
    check(lib().gfx_fir_spectrum_ex_f32(_ptr(h), _ptr(gain), gain_div, _ptr(Hs), RCf, N, part_len, _stream()),
          "gfx_fir_spectrum_ex_f32")
    return Hs


@_on_device
def fir_spectrum_reversed(x, part_len=0):
    """Spectra of the time-reversed rows of ``x`` ((R,C,L) or a strided (B,n,C,L) view, read in place): what
    ``fir_spectrum(x.flip(-1).reshape(R * C, L), part_len=part_len)`` returns, without the flipped copy."""
    _require_gpu(x)
    xmap, R, C, L = rowmap(x)
    Hs = torch.empty(lib().gfx_fir_spectrum_bytes_ex(R * C, L, part_len), dtype=torch.uint8, device=x.device)
    check(lib().gfx_fir_spectrum_rev_f32(_ptr(x), xmap, R, C, L, part_len, _ptr(Hs), _stream()),
          "gfx_fir_spectrum_rev_f32")
    return Hs


@_on_device
def fir_grad(x, g, N, off):
    """gh[r,c,k] = sum_n g[r,cg,n] x[r,cx,n+off-k], k < N <= 8193: the filter gradient of a short-filter convolution
    in one pass over x and g ((R,C,L) tensors or strided (B,n,C,L) views) -> (R, max(Cx,Cg), N)."""
    _require_gpu(x, g)
    xmap, R, Cx, L = rowmap(x)
    gmap, Rg, Cg, Lg = rowmap(g)
    if Rg != R:
        raise ValueError(f"fir_grad: {Rg} gradient rows for {R} signal rows")
    gh = torch.empty((R, max(Cx, Cg), N), dtype=torch.float32, device=x.device)
    with _timed("corr1_kernel", 4 * R * (Cx * L + Cg * Lg)):
        check(lib().gfx_fir_grad_f32(_ptr(x), xmap, _ptr(g), gmap, _ptr(gh), R, Cx, Cg, L, Lg, N, off, _stream()),
              "gfx_fir_grad_f32")
    return gh


def fftconv_can_tee(Cin, Cf, L, Lout, off, N):
    """Whether :func:`fftconv` can also write a copy of its input (gfx_fftconv_tee_f32's conditions)."""
    return off == 0 and Lout >= L and Cin >= Cf and lib().gfx_fftconv_nparts(N) == 1


GFX_EINVAL = -1                                  # include/grafx_amd.h
SCHEDULES = {"auto": 0, "tile": 1, "pipe": 2}   # GFX_SCHED_* of include/grafx_amd.h
# What `schedule="auto"` means to fftconv(): "auto" (the library decides: the persistent hand-scheduled kernel for large
# launches it covers) or "pipe" (prefer that kernel at every size it covers -- tests and latency experiments).
FFTCONV_SCHEDULE = "auto"
# the full-length convolution in front of the odd-length aliasing leaves the rows' maxima for its pair scaling (round 6;
# GRAFX_ROWMAX_BYPRODUCT=0: the aliasing takes them in a pass of its own over z, as for every other producer of z)
ROWMAX_BYPRODUCT = os.environ.get("GRAFX_ROWMAX_BYPRODUCT", "1") != "0"


@_on_device
def fftconv(x, Hs, N, Cf, Lout=None, off=0, out=None, tee=None, h_rows=None, part_len=0, schedule="auto", rowmax=None):
    """y[r,c,n] = sum_k h[r % h_rows,cf,k] x[r,cx,n+off-k], n < Lout (x zero outside [0,L)).

    ``x`` / ``out`` may be (R,C,L) tensors or strided (B,n,C,L) views (see :func:`rowmap`).
    ``tee``: optional tensor shaped like ``x`` that receives a copy of ``x`` from the same kernel.
    ``h_rows``: number of filters in ``Hs`` when fewer than the signal rows (rows are batch-major, so
    ``h_rows = nodes`` shares one filter per node across the batch); default: one filter per row.
    ``schedule``: "auto" (the library picks), "tile" (one tile per workgroup, compiler-scheduled) or "pipe" (the
    hand-scheduled persistent kernel, N <= 8193); see gfx_fftconv_sched_f32.
    ``rowmax`` (with schedule "auto"): a dict that receives ``rowmax["words"]`` -- an int32 tensor of R * max(C, Cf) words,
    the bits of max |y| of every output row-channel -- when the kernel that ran leaves them as a by-product
    (gfx_fftconv_rowmax_f32: the compiler-built tile kernels, one partition or many); untouched otherwise.  For odd_alias(rowmax=).
    """
    _require_gpu(x, out, tee)
    xmap, R, Cin, L = rowmap(x)
    Lout = L if Lout is None else Lout
    Cout = max(Cin, Cf)
    h_rows = R if h_rows is None else h_rows
    if out is None:
        out = torch.empty((R, Cout, Lout), dtype=torch.float32, device=x.device)
    ymap, Ry, Cy, Ly = rowmap(out)
    if (Ry, Cy) != (R, Cout) or Ly < Lout:
        raise ValueError(f"output shape {tuple(out.shape)} does not match rows={R}, channels={Cout}, length>={Lout}")
    if Hs.numel() != lib().gfx_fir_spectrum_bytes_ex(h_rows * Cf, N, part_len):
        raise ValueError(f"filter spectra hold {Hs.numel()} bytes, expected {h_rows} x {Cf} filters of {N} taps")
    nbytes = lib().gfx_fftconv_workspace_bytes_ex(R, Cin, L, Lout, off, N, part_len)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    cmap = RowMap(1, 0, 0, 0)
    if tee is not None:
        cmap, Rc, Cc, Lc = rowmap(tee)
        if (Rc, Cc, Lc) != (R, Cin, L):
            raise ValueError(f"tee shape {tuple(tee.shape)} does not match the input {tuple(x.shape)}")
    def launch(sched):
        return lib().gfx_fftconv_sched_f32(_ptr(x), xmap, _ptr(Hs), h_rows, part_len, _ptr(out), ymap, _ptr(tee), cmap, R, Cin,
                                           Cf, L, Lout, off, N, _ptr(ws), nbytes, SCHEDULES[sched], _stream())

    if rowmax is not None and schedule == "auto" and FFTCONV_SCHEDULE == "auto" and ROWMAX_BYPRODUCT:
        words = torch.zeros(R * Cout, dtype=torch.int32, device=x.device)
        written = ctypes.c_int(0)
        with _timed("fftconv", 4 * R * ((2 if tee is not None else 1) * Cin * L + Cout * Lout)) as t:
            check(lib().gfx_fftconv_rowmax_f32(_ptr(x), xmap, _ptr(Hs), h_rows, part_len, _ptr(out), ymap, _ptr(tee), cmap, R,
                                               Cin, Cf, L, Lout, off, N, _ptr(ws), nbytes, words.data_ptr(),
                                               ctypes.byref(written), _stream()), "gfx_fftconv_rowmax_f32")
            if t.rec is not None:
                t.name = lib().gfx_fftconv_last_kernel().decode()
        if written.value:
            rowmax["words"] = words
        return out
    with _timed("fftconv", 4 * R * ((2 if tee is not None else 1) * Cin * L + Cout * Lout)) as t:
        rc = GFX_EINVAL
        if schedule == "auto" and FFTCONV_SCHEDULE == "pipe":
            rc = launch("pipe")
            if rc not in (0, GFX_EINVAL):  # only "not covered by the persistent kernel" falls back to the library's choice;
                check(rc, "gfx_fftconv_sched_f32 (pipe)")   # a failed launch / code-object load must not be masked
        if rc != 0:
            check(launch(schedule), "gfx_fftconv_sched_f32")
        if t.rec is not None:             # the record is keyed by the kernel's own name, as a profile prints it
            t.name = lib().gfx_fftconv_last_kernel().decode()
    return out


@_on_device
def fir_direct(x, h, Lout=None, off=0, out=None, h_rows=None):
    """Short FIR (N <= 512 taps) on the fp32 matrix cores, taps given directly: same result as
    ``fftconv(x, fir_spectrum(h.reshape(-1, N)), N, Cf, Lout, off)``.  ``h``: (h_rows, Cf, N)."""
    _require_gpu(x, h, out)
    xmap, R, Cin, L = rowmap(x)
    h = h.contiguous()
    hr, Cf, N = h.shape
    h_rows = hr if h_rows is None else h_rows
    if hr != h_rows:
        raise ValueError(f"fir_direct: {hr} filter rows given, h_rows={h_rows}")
    Lout = L if Lout is None else Lout
    Cout = max(Cin, Cf)
    if out is None:
        out = torch.empty((R, Cout, Lout), dtype=torch.float32, device=x.device)
    ymap, Ry, Cy, Ly = rowmap(out)
    if (Ry, Cy) != (R, Cout) or Ly < Lout:
        raise ValueError(f"output shape {tuple(out.shape)} does not match rows={R}, channels={Cout}, length>={Lout}")
    with _timed("fir_mfma_kernel", 4 * R * (Cin * L + Cout * Lout)):
        check(lib().gfx_fir_direct_f32(_ptr(x), xmap, _ptr(h), h_rows, _ptr(out), ymap, R, Cin, Cf, L, Lout, off, N, _stream()),
              "gfx_fir_direct_f32")
    return out


# ----------------------------------------------------------------------------------------- small inverse real DFT
IRDFT_MAX_N = 8192


class FftLibraryWarning(UserWarning):
    """A parameter-sized transform left the library's own kernels for torch.fft (rocFFT): see fft_library_reached."""


def fft_library_reached(what):
    """The few parameter-sized transforms that are longer than the direct-sum / tile kernels cover (DESIGN.md section 8:
    never a signal, never on a BASELINE configuration) go to torch.fft -- and say so, once per call site."""
    import warnings

    warnings.warn(f"{what}: beyond the native transform sizes, computed by torch.fft (rocFFT)", FftLibraryWarning, stacklevel=3)


@_on_device
def irdft(X, n, roll=0, window=None):
    """irfft(X, n) for short transforms (n <= 8192, any n) as a direct sum on the GPU, rolled by ``roll`` and windowed:
    X (..., n//2 + 1) real or complex64 -> (..., n) float32 (gfx_irdft_f32)."""
    _require_gpu(window)
    if not X.is_cuda:
        _require_gpu(X)
    K = X.shape[-1]
    is_real = not X.is_complex()
    if X.dtype not in (torch.float32, torch.complex64):
        raise TypeError(f"irdft: float32 / complex64 input expected, got {X.dtype}")
    if window is not None and (window.dtype != torch.float32 or window.numel() != n):
        raise ValueError(f"irdft: the window must hold {n} float32 values, got {tuple(window.shape)} {window.dtype}")
    flat = (X if is_real else torch.view_as_real(X.contiguous())).reshape(-1, K, *(() if is_real else (2,))).contiguous()
    rows = flat.shape[0]
    y = torch.empty((rows, n), dtype=torch.float32, device=X.device)
    if rows == 0:        # an empty shard (gather_outputs / shard_batch produce them): torch.fft.irfft returned empty too
        return y.view(*X.shape[:-1], n)
    pin = _Pin()
    check(lib().gfx_irdft_f32(_ptr(flat), int(is_real), _ptr(y), rows, K, n, roll, pin(window), _stream()), "gfx_irdft_f32")
    return y.view(*X.shape[:-1], n)


@_on_device
def rdft(x, n=None):
    """rfft(x, n) along the last axis for short transforms (n <= 8192, any n) as a direct sum on the GPU (gfx_rdft_f32):
    x (..., n) float32 -> (..., n//2 + 1) complex64."""
    _require_gpu(x)
    n = x.shape[-1] if n is None else n
    if x.shape[-1] != n:
        raise ValueError(f"rdft: {x.shape[-1]} samples per row, n = {n}")
    K = n // 2 + 1
    flat = x.reshape(-1, n).contiguous()
    rows = flat.shape[0]
    X = torch.empty((rows, K, 2), dtype=torch.float32, device=x.device)
    for i in range(0, rows, 65535):     # rows ride on a grid dimension
        m = min(65535, rows - i)
        check(lib().gfx_rdft_f32(_ptr(flat[i:]), _ptr(X[i:]), m, K, n, _stream()), "gfx_rdft_f32")
    return torch.view_as_complex(X).view(*x.shape[:-1], K)


# ----------------------------------------------------------------------------------------- odd-length aliasing
_ALIAS_PLANS = {}          # (P, device) -> plan, most recently used last; at most _ALIAS_PLANS_MAX entries (up to 36 MB each, 72 MB precise)
_ALIAS_PLANS_MAX = max(1, int(os.environ.get("GRAFX_ALIAS_PLANS", "8")))


def set_alias_plan_cache(entries):
    """How many chirp plans of the odd-length aliasing (one per distinct P = Lx + Lh - 1 and precision, up to 36 / 72 MB
    each) stay cached per process; default 8, or GRAFX_ALIAS_PLANS.  A workload with many distinct lengths (ragged
    batches through upstream's default even tap counts) rebuilds a plan -- one small kernel chain and a stream
    synchronisation -- whenever a length falls out of the cache: size it to the number of distinct lengths in flight."""
    global _ALIAS_PLANS_MAX
    _ALIAS_PLANS_MAX = max(1, int(entries))



def odd_alias_supported(P):
    """Whether gfx_odd_alias_f32 handles a linear convolution of odd length P (3 <= P <= 11,184,811)."""
    return lib().gfx_odd_alias_plan_bytes(P) > 0


# Two real rows per complex chirp-z transform (csrc/czt_pair.hip) in odd_alias's forward calls: a third fewer bytes
# through every pass.  GRAFX_ALIAS_PAIRS=0 keeps one transform per row (round 4's path; also what rows beyond
# P = 8 388 607 and every adjoint take).
ALIAS_PAIRS = os.environ.get("GRAFX_ALIAS_PAIRS", "1") != "0"
ALIAS_ROWS_PER_CHUNK = int(os.environ.get("GRAFX_ALIAS_ROWS", "1024"))   # rows of one launch chain (and ALIAS_WS_CAP bytes at most)


def _alias_fns(precise, pairs=False):
    """(plan_bytes, workspace_bytes, plan, forward, adjoint, tag) of the fp32 or the double-precision transforms; ``pairs``:
    the two-rows-per-transform forms (forward only: adjoint is None)."""
    L, tag = lib(), "gfx_odd_alias_" + ("pair_" if pairs else "") + ("precise_" if precise else "")
    return (getattr(L, tag + "plan_bytes"), getattr(L, tag + "workspace_bytes"), getattr(L, tag + "plan_f32"),
            getattr(L, tag + "f32"), None if pairs else getattr(L, tag + "adjoint_f32"), tag)


def _alias_pairs(P, rows):
    """Whether a forward call of `rows` rows of length P goes two rows per transform."""
    return ALIAS_PAIRS and rows >= 2 and lib().gfx_odd_alias_pair_plan_bytes(P) > 0


def _alias_plan(P, device, precise=False, pairs=False):
    """The per-P chirp plan of the aliasing kernels (LRU of _ALIAS_PLANS_MAX, built on first use)."""
    plan_bytes, ws_bytes, build, _, _, tag = _alias_fns(precise, pairs)
    key = (P, device.type, device.index, bool(precise), bool(pairs))
    plan = _ALIAS_PLANS.pop(key, None)
    if plan is not None:
        _ALIAS_PLANS[key] = plan   # back in as the most recent
    if plan is None:
        if torch.cuda.is_current_stream_capturing():
            # a plan built during capture would live in the graph's private pool and be rebuilt on every replay
            raise RuntimeError(f"odd_alias: no plan for P={P} yet; run the call once outside the HIP-graph capture")
        plan = torch.empty(plan_bytes(P), dtype=torch.uint8, device=device)
        w1 = torch.empty(ws_bytes(1, P), dtype=torch.uint8, device=device)
        check(build(_ptr(plan), P, _ptr(w1), w1.numel(), _stream()), tag + "plan_f32")
        torch.cuda.current_stream(device).synchronize()   # the cached plan is complete before any stream can pick it up
        _ALIAS_PLANS[key] = plan
        while len(_ALIAS_PLANS) > _ALIAS_PLANS_MAX:
            old = _ALIAS_PLANS.pop(next(iter(_ALIAS_PLANS)))
            torch.cuda.synchronize(old.device)               # nobody on any stream still reads the evicted plan
            del old
    plan.record_stream(torch.cuda.current_stream(device))
    return plan


ALIAS_WS_CAP = 4 << 30     # bytes of chirp-z workspace per launch chain (rows go through in chunks that fit it; 1 -> 4 GB:
                           # -6 % on the console's 4608 equaliser rows, tools/czt_chunk_ab.py).  Never more than a quarter
                           # of the memory that is free when the call is made, and halved (down to one row) when the
                           # allocation fails: a workload that fitted with the 1 GB chains of round 3 still fits.


def set_alias_workspace_cap(nbytes):
    """Upper bound, in bytes, of the transient chirp-z workspace one odd_alias / odd_alias_adjoint launch chain may hold
    (default 4 GiB; the effective bound is also a quarter of the free device memory).  Returns the previous value."""
    global ALIAS_WS_CAP
    old, ALIAS_WS_CAP = ALIAS_WS_CAP, max(int(nbytes), 1)
    return old


def _alias_chunks(rows, P, rows_per_chunk, device, precise, pairs=False):
    ws_bytes = _alias_fns(precise, pairs)[1]
    cap = ALIAS_WS_CAP
    if not torch.cuda.is_current_stream_capturing():
        # free to this process = what the driver has left + what torch's caching allocator holds without using it
        # (in a long-running / training process the cache holds most of the device: the driver's figure alone would
        # collapse the chunks to a few rows)
        free = torch.cuda.mem_get_info(device)[0] + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        cap = min(cap, max(free // 4, 1))
    unit = 2 if pairs else 1            # chunks of whole pairs: row 2r stays with row 2r + 1
    chunk = max(unit, min(rows, rows_per_chunk, unit * (cap // ws_bytes(unit, P))) // unit * unit)
    if chunk >= rows:
        chunk = rows
    while True:
        try:
            return chunk, torch.empty(ws_bytes(chunk, P), dtype=torch.uint8, device=device)
        except torch.OutOfMemoryError:
            if chunk <= unit:
                raise
            chunk = max(unit, chunk // 2 // unit * unit)


@_on_device
def odd_alias(z, lo=0, length=None, rows_per_chunk=None, precise=False, out=None, rowmax=None, relu=False):
    """irfft_{P-1}(rfft_P(z))[..., lo : lo + length] for z (..., P), P odd: the reference convolve()'s aliasing of a
    full linear convolution (core/convolution.py:123-126), on the chirp-z kernels.  Rows go through in chunks (1.6 MB of
    workspace per row at P ~ 135 k: 25 tiles of 8192 points).  ``precise``: transforms in double precision (twice the
    workspace), for results that feed a logarithm -- the energy envelope, core/envelope.py:34-49.
    ``rowmax``: int32 words, one per row of z, holding the bits of max |z| of the row (what fftconv(rowmax=) leaves): the
    two-rows-per-transform form then skips its own pass over z.  ``relu`` (with ``precise`` and pairs): the last pass stores
    max(y, 0) -- the envelope smoother's clamp (core/envelope.py:48) without a pass of its own."""
    _require_gpu(z)
    P = z.shape[-1]
    Q = P - 1
    length = Q - lo if length is None else length
    flat = z.reshape(-1, P).contiguous()
    rows = flat.shape[0]
    rows_per_chunk = ALIAS_ROWS_PER_CHUNK if rows_per_chunk is None else rows_per_chunk
    pairs = _alias_pairs(P, rows)
    plan = _alias_plan(P, z.device, precise, pairs)
    fwd, tag = _alias_fns(precise, pairs)[3], _alias_fns(precise, pairs)[5]
    chunk, ws = _alias_chunks(rows, P, rows_per_chunk, z.device, precise, pairs)
    if rowmax is not None and (not pairs or rowmax.numel() != rows or rowmax.dtype != torch.int32 or not rowmax.is_cuda):
        rowmax = None
    if relu and not (pairs and precise and out is None):
        raise ValueError("odd_alias: relu is fused into the two-rows-per-transform double-precision form only")
    if out is not None:
        # ``out``: a (R, C, length) tensor or a strided (B, n, C, length) view whose rows, channels flattened, are z's
        # rows: the last column pass writes them in place (gfx_odd_alias_rows_f32; float transforms only)
        _require_gpu(out)
        omap, Ro, Co, Lo = rowmap(out)
        if precise or Ro * Co != rows or Lo != length:
            raise ValueError(f"odd_alias: out {tuple(out.shape)} does not take {rows} rows of {length} samples")
        name = ("czt_pair_in_kernel+czt_rows_kernel+czt_pair_mid_kernel+czt_pair_out_kernel<float>" if pairs
                else "czt_cols_fwd_kernel+czt_rows_kernel+czt_cols_inv_kernel<float>")
        rows_fn = lib().gfx_odd_alias_pair_rows_f32 if pairs else lib().gfx_odd_alias_rows_f32
        for i in range(0, rows, chunk):
            n = min(chunk, rows - i)
            with _timed(name, 4 * n * (P + length)):
                if rowmax is not None:
                    check(lib().gfx_odd_alias_pair_rows_max_f32(_ptr(flat[i : i + n]), _ptr(out), omap, Co, i, lo, length, n, P,
                                                                _ptr(plan), _ptr(ws), ws.numel(), rowmax[i : i + n].data_ptr(),
                                                                _stream()), "gfx_odd_alias_pair_rows_max_f32")
                else:
                    check(rows_fn(_ptr(flat[i : i + n]), _ptr(out), omap, Co, i, lo, length, n, P, _ptr(plan), _ptr(ws),
                                  ws.numel(), _stream()), tag + "rows_f32")
        return out
    out = torch.empty((rows, length), dtype=torch.float32, device=z.device)
    # one record per chunk: the column / tile / column passes of the two chirp-z transforms (czt.hip), read z + write y
    name = (("czt_pair_in_kernel+czt_rows_kernel+czt_pair_mid_kernel+czt_pair_out_kernel" if pairs
             else "czt_cols_fwd_kernel+czt_rows_kernel+czt_cols_inv_kernel") + ("<double>" if precise else "<float>"))
    for i in range(0, rows, chunk):
        n = min(chunk, rows - i)
        with _timed(name, 4 * n * (P + length)):
            if precise and pairs and (rowmax is not None or relu):
                check(lib().gfx_odd_alias_pair_precise_max_f32(_ptr(flat[i : i + n]), _ptr(out[i : i + n]), length, lo, length, n, P,
                                                               _ptr(plan), _ptr(ws), ws.numel(),
                                                               None if rowmax is None else rowmax[i : i + n].data_ptr(),
                                                               int(relu), _stream()), "gfx_odd_alias_pair_precise_max_f32")
            elif rowmax is not None:
                check(lib().gfx_odd_alias_pair_max_f32(_ptr(flat[i : i + n]), _ptr(out[i : i + n]), length, lo, length, n, P,
                                                       _ptr(plan), _ptr(ws), ws.numel(), rowmax[i : i + n].data_ptr(), _stream()),
                      "gfx_odd_alias_pair_max_f32")
            else:
                check(fwd(_ptr(flat[i : i + n]), _ptr(out[i : i + n]), length, lo, length, n, P, _ptr(plan), _ptr(ws), ws.numel(),
                          _stream()), tag + "f32")
    return out.view(*z.shape[:-1], length)


@_on_device
def odd_alias_adjoint(gy, P, lo=0, rows_per_chunk=1024, precise=False):
    """Transpose of odd_alias: the gradient with respect to z (..., P) given gy (..., length), the gradient with respect
    to odd_alias(z, lo, length) -- what autograd derives from the reference's rfft / irfft pair."""
    _require_gpu(gy)
    length = gy.shape[-1]
    plan = _alias_plan(P, gy.device, precise)
    adj, tag = _alias_fns(precise)[4], _alias_fns(precise)[5]
    flat = gy.reshape(-1, length).contiguous()
    rows = flat.shape[0]
    out = torch.empty((rows, P), dtype=torch.float32, device=gy.device)
    chunk, ws = _alias_chunks(rows, P, rows_per_chunk, gy.device, precise)
    for i in range(0, rows, chunk):
        n = min(chunk, rows - i)
        check(adj(_ptr(flat[i : i + n]), length, lo, length, _ptr(out[i : i + n]), n, P, _ptr(plan), _ptr(ws), ws.numel(),
                  _stream()), tag + "adjoint_f32")
    return out.view(*gy.shape[:-1], P)


# ----------------------------------------------------------------------------------------- IIR (FSM)
@_on_device
def iir_fsm_native(N):
    """Whether the library has a native tap kernel for fsm_fir_len = N (1..4096, 8192, 16384)."""
    return bool(lib().gfx_iir_fsm_native(N))


def iir_fsm_plan(N, device):
    nbytes = lib().gfx_iir_fsm_plan_bytes(N)
    if nbytes == 0:
        if iir_fsm_native(N):
            return None  # power-of-two lengths above the Bluestein limit need no plan
        raise NotImplementedError(f"fsm_fir_len={N}: the HIP FSM kernels support 1 <= fsm_fir_len <= 4096, 8192 and 16384")
    plan = torch.empty(nbytes, dtype=torch.uint8, device=device)
    check(lib().gfx_iir_fsm_plan_f32(_ptr(plan), N, _stream()), "gfx_iir_fsm_plan_f32")
    return plan


@_on_device
def iir_fsm_fir(Bs, As, N, plan):
    """(R, Cf, K, 3) biquad coefficients -> (R*Cf, N) frequency-sampled FIR taps.  float64 coefficients take the
    double-precision response evaluation (gfx_iir_fsm_fir_f64c_f32); the taps are float32 either way."""
    if Bs.dtype == torch.float64 and As.dtype == torch.float64:
        if not (Bs.is_cuda and As.is_cuda):
            _require_gpu(Bs.float(), As.float())
    else:
        _require_gpu(Bs, As)
    Bs, As = Bs.contiguous(), As.contiguous()
    K = Bs.shape[-2]
    RC = Bs.numel() // (K * 3)
    h = torch.empty((RC, N), dtype=torch.float32, device=Bs.device)
    if Bs.dtype == torch.float64:   # coefficients carried in double precision (gfx_iir_fsm_fir_f64c_f32): float32 taps
        if As.dtype != torch.float64:
            raise TypeError("iir_fsm_fir: Bs and As must have the same dtype")
        check(lib().gfx_iir_fsm_fir_f64c_f32(Bs.data_ptr(), As.data_ptr(), _ptr(plan), _ptr(h), RC, K, N, _stream()),
              "gfx_iir_fsm_fir_f64c_f32")
        return h
    check(lib().gfx_iir_fsm_fir_f32(_ptr(Bs), _ptr(As), _ptr(plan), _ptr(h), RC, K, N, _stream()), "gfx_iir_fsm_fir_f32")
    return h


@_on_device
def iir_fsm_bwd(Bs, As, G, delays, N, want_B=True, want_A=True):
    """Gradient of iir_fsm_fir's taps with respect to (R, Cf, K, 3) float32 coefficients from G = rfft(dL/dh, n = N)
    ((R, Cf, N // 2 + 1) complex64) and the forward pass's (3, N // 2 + 1) complex64 delay table: one native launch in double
    precision (gfx_iir_fsm_bwd_f32) -> (gB, gA), None for the one not wanted."""
    _require_gpu(Bs, As)
    if not (G.is_cuda and delays.is_cuda):
        _require_gpu(torch.view_as_real(G), torch.view_as_real(delays))
    Bs, As = Bs.contiguous(), As.contiguous()
    K = Bs.shape[-2]
    RC = Bs.numel() // (K * 3)
    F = N // 2 + 1
    if G.dtype != torch.complex64 or delays.dtype != torch.complex64 or G.numel() != RC * F or tuple(delays.shape) != (3, F):
        raise ValueError(f"iir_fsm_bwd: G {tuple(G.shape)} {G.dtype} / delays {tuple(delays.shape)} {delays.dtype} do not "
                         f"match {RC} filters of {N} taps")
    Gr, Dr = torch.view_as_real(G.contiguous()), torch.view_as_real(delays.contiguous())
    gB = torch.empty_like(Bs) if want_B else None
    gA = torch.empty_like(As) if want_A else None
    check(lib().gfx_iir_fsm_bwd_f32(_ptr(Bs), _ptr(As), _ptr(Gr), _ptr(Dr), _ptr(gB), _ptr(gA), RC, K, N, _stream()),
          "gfx_iir_fsm_bwd_f32")
    return gB, gA


@_on_device
def peq_coeffs(w0, q_inv, log_gain, use_shelving=True):
    _require_gpu(w0, q_inv, log_gain)
    w0, q_inv, log_gain = w0.contiguous(), q_inv.contiguous(), log_gain.contiguous()
    K = w0.shape[-1]
    n = w0.numel() // K
    Bs = torch.empty((*w0.shape, 3), dtype=torch.float32, device=w0.device)
    As = torch.empty_like(Bs)
    check(lib().gfx_peq_coeffs_f32(_ptr(w0), _ptr(q_inv), _ptr(log_gain), _ptr(Bs), _ptr(As), n, K, int(use_shelving), _stream()),
          "gfx_peq_coeffs_f32")
    return Bs, As


@_on_device
def peq_coeffs_bwd(w0, q_inv, log_gain, gBs, gAs, use_shelving=True):
    """Gradient of :func:`peq_coeffs` -> (g_w0, g_q_inv, g_log_gain), each shaped like w0."""
    _require_gpu(w0, q_inv, log_gain, gBs, gAs)
    w0, q_inv, log_gain = w0.contiguous(), q_inv.contiguous(), log_gain.contiguous()
    _expect(gBs, (*w0.shape, 3), "peq_coeffs_bwd: gBs")
    _expect(gAs, (*w0.shape, 3), "peq_coeffs_bwd: gAs")
    K = w0.shape[-1]
    out = [torch.empty_like(w0) for _ in range(3)]
    pin = _Pin()
    check(lib().gfx_peq_coeffs_bwd_f32(_ptr(w0), _ptr(q_inv), _ptr(log_gain), pin(gBs), pin(gAs),
                                       *(_ptr(o) for o in out), w0.numel() // K, K, int(use_shelving), _stream()),
          "gfx_peq_coeffs_bwd_f32")
    return tuple(out)


@_on_device
def biquad_coeffs(Bs_in, A1_pre, A2_pre, A0=None):
    _require_gpu(Bs_in, A1_pre, A2_pre, A0)
    Bs_in, A1_pre, A2_pre = Bs_in.contiguous(), A1_pre.contiguous(), A2_pre.contiguous()
    A0 = None if A0 is None else A0.contiguous()
    Bs = torch.empty_like(Bs_in)
    As = torch.empty_like(Bs_in)
    check(lib().gfx_biquad_coeffs_f32(_ptr(Bs_in), _ptr(A1_pre), _ptr(A2_pre), _ptr(A0), _ptr(Bs), _ptr(As), A1_pre.numel(), _stream()),
          "gfx_biquad_coeffs_f32")
    return Bs, As


# ----------------------------------------------------------------------------------------- dynamics
KNEES = {"hard": 0, "quadratic": 1, "exponential": 2}


def _rowvec(p, R):
    if p is None:
        return None
    _require_gpu(p)
    p = p.contiguous()
    if p.numel() != R:
        raise ValueError(f"per-row parameter has {p.numel()} elements for {R} rows")
    return p


# Default schedule of dynamics_fused: "oneshot" hands the library a workspace, with which the smoothed configuration runs
# as dependency-free 1024-sample tiles for every row whose smoother memory is short (decided per row on the device,
# gfx_dynamics_fused_ws_f32) and as one workgroup per row for the others; "rows" forces one workgroup per row.
MIX_FUSION = True          # dynamics stages take the routing sum that follows them (see dynamics_fused(mix=))
DYN_SCHEDULE = "oneshot"
# the compressor backward without a kept scan rebuilds it inside its tiles (gfx_dynamics_bwd_rescan_ws_f32); False: a pass
# over every row writes it out first (gfx_dynamics_bwd_f32, rounds 2-5)
DYN_BWD_RESCAN = os.environ.get("GRAFX_DYN_BWD_RESCAN", "1") != "0"
# rows with a long smoother memory stay on the tile grid (gfx_dynamics_ws_bytes_ex); False / GRAFX_DYN_LOOKBACK=0: round 4
DYN_LOOKBACK = os.environ.get("GRAFX_DYN_LOOKBACK", "1") != "0"


def mix_schedule(dest_sources, n):
    """The schedule of gfx_dynamics_fused_mix_f32 for a routing sum over the n rows of a graph: ``dest_sources[d]`` = the
    rows (strictly increasing) added into destination d, counted from the stage's first row -- 0 .. n-1 are the stage's
    own rows, negative numbers and numbers >= n are finished rows of the buffer before / behind them ("extras").  A
    destination occupies an accumulator from its first to its last source; live ranges are coloured greedily.
    -> (codes (list of n ints), accumulators used, pre, post) with pre / post = [(row, code), ...] for the extras in
    front of / behind the stage's rows; or None when a destination has no source, its rows are not increasing, no
    destination takes any of the stage's rows, or more than four destinations are live at once."""
    if not dest_sources or len(dest_sources) > 254:
        return None
    for rows in dest_sources:
        if not rows or any(b <= a for a, b in zip(rows, rows[1:])):
            return None
    if not any(0 <= r < n for rows in dest_sources for r in rows):
        return None
    pre = sorted({r for rows in dest_sources for r in rows if r < 0})
    post = sorted({r for rows in dest_sources for r in rows if r >= n})
    pos = {r: k for k, r in enumerate(pre)}
    pos.update({j: len(pre) + j for j in range(n)})
    pos.update({r: len(pre) + n + k for k, r in enumerate(post)})
    codes, free_at, slot = [0] * (len(pre) + n + len(post)), [0] * 4, {}
    for d in sorted(range(len(dest_sources)), key=lambda d: pos[dest_sources[d][0]]):
        first, last = pos[dest_sources[d][0]], pos[dest_sources[d][-1]]
        a = next((k for k in range(4) if free_at[k] <= first), None)
        if a is None:
            return None
        free_at[a], slot[d] = last + 1, a
        for r in dest_sources[d]:
            codes[pos[r]] |= 1 << a
        codes[last] |= (d + 1) << (8 + 8 * a)
    return (codes[len(pre) : len(pre) + n], max(slot.values()) + 1, list(zip(pre, codes[: len(pre)])),
            list(zip(post, codes[len(pre) + n :])))


@_on_device
def dynamics_fused(x, log_threshold, log_ratio, log_knee, z_alpha, smoother, iir_len, knee, gate, out=None,
                   param_rows=None, schedule=None, u1_out=None, mix=None):
    """``param_rows``: number of parameter rows when shared across the batch (row r uses r % param_rows).
    ``schedule``: "oneshot" (default) or "rows", see above.
    ``u1_out``: optional (R, L) tensor that receives the smoother's un-truncated scan for :func:`dynamics_bwd` (the
    training forward; smoother = 1 only).
    ``mix``: a dict {"sched": int64 device tensor (n,), "n_acc": int, "out": (B, J, C, L) view, optionally "extras": int64
    device tensor (n_pre + n_post, 2) of (row offset from ``out``'s first row, code) and "n_pre"} for a strided
    (B, n, C, L) input (see :func:`mix_schedule`) -- the routing sum that follows is computed by the same kernel
    (gfx_dynamics_fused_mix_f32) and ``mix["done"]`` is set; when the configuration cannot take it, nothing is summed and
    ``mix["done"]`` stays unset (the caller runs the gather-sum stage)."""
    schedule = DYN_SCHEDULE if schedule is None else schedule
    if schedule not in ("oneshot", "rows"):
        raise ValueError(f"dynamics_fused: unknown schedule {schedule!r}")
    _require_gpu(x, out)
    xmap, R, C, L = rowmap(x)
    P = R if param_rows is None else param_rows
    if out is None:
        out = torch.empty((R, C, L), dtype=torch.float32, device=x.device)
    ymap = rowmap(out)[0]
    pin = _Pin()
    if u1_out is not None:
        _require_gpu(u1_out)
        _expect(u1_out, (R, L), "dynamics_fused: u1_out")
        if not u1_out.is_contiguous() or smoother != 1:
            raise ValueError("dynamics_fused: u1_out must be contiguous and needs the one-pole smoother")
    ws = None
    if smoother == 1 and schedule == "oneshot":
        ws = torch.empty(lib().gfx_dynamics_ws_bytes_ex(P, R, L) if DYN_LOOKBACK else lib().gfx_dynamics_ws_bytes(P),
                         dtype=torch.uint8, device=x.device)
    args = (_ptr(x), xmap, _ptr(out), ymap, pin(_rowvec(log_threshold, P)), pin(_rowvec(log_ratio, P)),
            pin(_rowvec(log_knee, P)), pin(_rowvec(z_alpha, P)), P, R, C, L, smoother, iir_len, KNEES[knee], int(gate),
            _ptr(u1_out), _ptr(ws), 0 if ws is None else ws.numel(), _stream())
    if mix is not None and ws is not None and MIX_FUSION and x.ndim == 4 and out.ndim == 4:
        mo, sched = mix["out"], mix["sched"]
        n, J = x.shape[1], mo.shape[1]
        if (mo.shape == (x.shape[0], J, C, L) and mo.stride(-1) == 1 and sched.numel() == n and xmap.inner == n
                and ymap.inner == n):
            own = 4 if mix.get("skip_rows") else 8
            with _timed("dyn_fused_kernel", own * R * C * L + 4 * mo.numel() + (4 * R * L if u1_out is not None else 0)) as t:
                ex = mix.get("extras")
                n_ex = 0 if ex is None else ex.shape[0]
                n_pre = mix.get("n_pre", 0)
                # mix["skip_rows"]: nobody but these sums reads the stage's rows (an output-only render): do not store them
                rc = lib().gfx_dynamics_fused_mix_flags_f32(*args[:-1], _ptr(sched), n, mix["n_acc"], _ptr(mo), mo.stride(0),
                                                            mo.stride(1), mo.stride(2) if C == 2 else 0, _ptr(ex), n_pre,
                                                            n_ex - n_pre, 1 if mix.get("skip_rows") else 0, _stream())
                if rc == 0 and t.rec is not None:    # keyed by the kernel's own name, as a profile prints it
                    t.name = lib().gfx_dynamics_last_kernel().decode()
            if rc == 0:
                mix["done"] = True
                return out
            if rc != -1:   # GFX_EINVAL: not a configuration of the fused kernel
                check(rc, "gfx_dynamics_fused_mix_f32")
    with _timed("dyn_fused_kernel", 8 * R * C * L + (4 * R * L if u1_out is not None else 0)) as t:
        check(lib().gfx_dynamics_fused_ws_f32(*args), "gfx_dynamics_fused_ws_f32")
        if t.rec is not None:
            t.name = lib().gfx_dynamics_last_kernel().decode()
    return out


@_on_device
def dyn_gain_bwd(x, gy, env, log_threshold, log_ratio, log_knee, knee, gate):
    """-> (gain (R,L), denv (R,L), gparams (R,3) = d/d(log_threshold, log_ratio, log_knee)); see the header."""
    _require_gpu(x, gy, env)
    xmap, R, C, L = rowmap(x)
    gmap, Rg, Cg, Lg = rowmap(gy)
    if (Rg, Cg, Lg) != (R, C, L):
        raise ValueError(f"dyn_gain_bwd: gradient {tuple(gy.shape)} does not match the input {tuple(x.shape)}")
    _expect(env, (R, L), "dyn_gain_bwd: env")
    env = env.contiguous()
    gain, denv = torch.empty_like(env), torch.empty_like(env)
    gp = torch.zeros((R, 3), dtype=torch.float32, device=x.device)
    pin = _Pin()
    check(lib().gfx_dyn_gain_bwd_f32(_ptr(x), xmap, _ptr(gy), gmap, _ptr(env), pin(_rowvec(log_threshold, R)),
                                     pin(_rowvec(log_ratio, R)), pin(_rowvec(log_knee, R)), R, C, L, KNEES[knee],
                                     int(gate), _ptr(gain), _ptr(denv), _ptr(gp), _stream()), "gfx_dyn_gain_bwd_f32")
    return gain, denv, gp


@_on_device
def dynamics_bwd(x, gy, log_threshold, log_ratio, log_knee, z_alpha, iir_len, knee, gate, out=None, pole=True, u1=None,
                 schedule=None, rescan=None):
    """Fused backward of the smoothed compressor / gate -> (gx (R,C,L), gparams (R,3), dalpha (R) or None).
    ``out``: optional destination for gx ((R,C,L) or a strided (B,n,C,L) view).  ``dalpha`` is the gradient with
    respect to the clamped pole a = min(sigmoid(z_alpha), 1 - 1e-5); the caller applies the chain rule to z_alpha.
    ``u1``: the scan kept by the forward pass (``dynamics_fused(..., u1_out=)``); without it the scan is recomputed --
    inside the backward tiles for rows with a short smoother memory (``rescan``, default DYN_BWD_RESCAN: the scratch the
    other rows' scan goes to is allocated but normally never touched), or by a pass of its own over every row.
    ``schedule`` (with ``u1``): "oneshot" (default, see DYN_SCHEDULE) or "rows"."""
    _require_gpu(x, gy, out)
    xmap, R, C, L = rowmap(x)
    gmap, Rg, Cg, Lg = rowmap(gy)
    if (Rg, Cg, Lg) != (R, C, L) or (out is not None and rowmap(out)[1:] != (R, C, L)):
        raise ValueError(f"dynamics_bwd: gradient / output shapes do not match the input {tuple(x.shape)}")
    gx = torch.empty((R, C, L), dtype=torch.float32, device=x.device) if out is None else out
    gp = torch.empty((R, 3), dtype=torch.float32, device=x.device)
    da = torch.empty(R, dtype=torch.float32, device=x.device) if pole else None
    pin = _Pin()
    if u1 is not None:
        _require_gpu(u1)
        _expect(u1, (R, L), "dynamics_bwd: u1")
        ws = None
        if (DYN_SCHEDULE if schedule is None else schedule) == "oneshot":   # one-shot tiles for the short-memory rows
            ws = torch.empty(lib().gfx_dynamics_bwd_ws_bytes(R, L), dtype=torch.uint8, device=x.device)
        check(lib().gfx_dynamics_bwd_u1_ws_f32(_ptr(x), xmap, _ptr(gy), gmap, pin(_rowvec(log_threshold, R)),
                                               pin(_rowvec(log_ratio, R)), pin(_rowvec(log_knee, R)),
                                               pin(_rowvec(z_alpha, R)), R, C, L, iir_len, KNEES[knee], int(gate), _ptr(gx),
                                               rowmap(gx)[0], _ptr(gp), pin(u1), _ptr(da), _ptr(ws),
                                               0 if ws is None else ws.numel(), _stream()), "gfx_dynamics_bwd_u1_ws_f32")
        return gx, gp, da
    u1 = torch.empty((R, L), dtype=torch.float32, device=x.device)
    if DYN_BWD_RESCAN if rescan is None else rescan:
        ws = torch.empty(lib().gfx_dynamics_bwd_ws_bytes(R, L), dtype=torch.uint8, device=x.device)
        check(lib().gfx_dynamics_bwd_rescan_ws_f32(_ptr(x), xmap, _ptr(gy), gmap, pin(_rowvec(log_threshold, R)),
                                                   pin(_rowvec(log_ratio, R)), pin(_rowvec(log_knee, R)),
                                                   pin(_rowvec(z_alpha, R)), R, C, L, iir_len, KNEES[knee], int(gate),
                                                   _ptr(gx), rowmap(gx)[0], _ptr(gp), _ptr(u1), _ptr(da), _ptr(ws), ws.numel(),
                                                   _stream()), "gfx_dynamics_bwd_rescan_ws_f32")
        return gx, gp, da
    check(lib().gfx_dynamics_bwd_f32(_ptr(x), xmap, _ptr(gy), gmap, pin(_rowvec(log_threshold, R)),
                                     pin(_rowvec(log_ratio, R)), pin(_rowvec(log_knee, R)), pin(_rowvec(z_alpha, R)),
                                     R, C, L, iir_len, KNEES[knee], int(gate), _ptr(gx), rowmap(gx)[0], _ptr(gp), None,
                                     _ptr(u1), _ptr(da), _stream()), "gfx_dynamics_bwd_f32")
    return gx, gp, da


@_on_device
def onepole_dz(g, U, D, coef, N):
    """Row sums sum_n g[n] (c0 U[n] + c2 U[n-N]) + g[n+1] (c1 D[n] + c3 D[n-N]); coef (R,4)."""
    _require_gpu(g, U, D, coef)
    R, L = g.shape
    _expect(U, (R, L), "onepole_dz: U")
    _expect(D, (R, L), "onepole_dz: D")
    _expect(coef, (R, 4), "onepole_dz: coef")
    da = torch.empty(R, dtype=torch.float32, device=g.device)
    pin = _Pin()
    check(lib().gfx_onepole_dz_f32(pin(g), pin(U), pin(D), pin(coef), _ptr(da), R, L, N, _stream()), "gfx_onepole_dz_f32")
    return da


@_on_device
def dyn_dx(x, gy, gain, de):
    _require_gpu(x, gy, gain, de)
    xmap, R, C, L = rowmap(x)
    if rowmap(gy)[1:] != (R, C, L):
        raise ValueError(f"dyn_dx: gradient {tuple(gy.shape)} does not match the input {tuple(x.shape)}")
    _expect(gain, (R, L), "dyn_dx: gain")
    _expect(de, (R, L), "dyn_dx: de")
    gx = torch.empty((R, C, L), dtype=torch.float32, device=x.device)
    pin = _Pin()
    check(lib().gfx_dyn_dx_f32(_ptr(x), xmap, _ptr(gy), rowmap(gy)[0], pin(gain), pin(de),
                               _ptr(gx), R, C, L, _stream()), "gfx_dyn_dx_f32")
    return gx


@_on_device
def energy(x):
    _require_gpu(x)
    xmap, R, C, L = rowmap(x)
    e = torch.empty((R, L), dtype=torch.float32, device=x.device)
    check(lib().gfx_energy_f32(_ptr(x), xmap, _ptr(e), R, C, L, _stream()), "gfx_energy_f32")
    return e


@_on_device
def onepole(u, z_alpha, iir_len, Lout=None, relu=True):
    _require_gpu(u, z_alpha)
    u = u.contiguous()
    R, L = u.shape
    Lout = L if Lout is None else Lout
    out = torch.empty((R, Lout), dtype=torch.float32, device=u.device)
    pin = _Pin()
    check(lib().gfx_onepole_f32(_ptr(u), pin(_rowvec(z_alpha, R)), _ptr(out), R, L, Lout, iir_len, int(relu), _stream()), "gfx_onepole_f32")
    return out


@_on_device
def onepole_energy(x, z_alpha, iir_len, Lout=None, relu=True, rowmax=None):
    """onepole(energy(x), ...) in one pass over the signal x ((R, C, L) or a strided (B, n, C, L) view): gfx_onepole_energy_f32.
    ``rowmax``: a dict that receives ``rowmax["words"]``, the bits of max |out| of every row (for odd_alias(rowmax=))."""
    _require_gpu(x, z_alpha)
    xmap, R, C, L = rowmap(x)
    Lout = L if Lout is None else Lout
    out = torch.empty((R, Lout), dtype=torch.float32, device=x.device)
    words = torch.empty(R, dtype=torch.int32, device=x.device) if rowmax is not None else None
    pin = _Pin()
    check(lib().gfx_onepole_energy_f32(_ptr(x), xmap, C, pin(_rowvec(z_alpha, R)), _ptr(out), R, L, Lout, iir_len, int(relu),
                                       None if words is None else words.data_ptr(), _stream()), "gfx_onepole_energy_f32")
    if rowmax is not None:
        rowmax["words"] = words
    return out


@_on_device
def onepole_fir(z_alpha, iir_len):
    _require_gpu(z_alpha)
    R = z_alpha.numel()
    h = torch.empty((R, iir_len), dtype=torch.float32, device=z_alpha.device)
    pin = _Pin()
    check(lib().gfx_onepole_fir_f32(pin(z_alpha), _ptr(h), R, iir_len, _stream()), "gfx_onepole_fir_f32")
    return h


BALLISTICS_SCHEDULE = "chunks"   # "chunks": rows cut into verified chunks (gfx_ballistics_ws_f32); "rows": whole rows only


def _overlap(a, b):
    """Whether two float32 tensors share an element.  Exact for views of one storage with the same shape and (nested)
    strides -- two node ranges of the render's (B, V, C, L) buffer interleave in memory without touching --, the bounding
    ranges otherwise."""
    if a.numel() == 0 or b.numel() == 0 or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
        return False
    if a.shape == b.shape and a.stride() == b.stride() and all(st > 0 for st in a.stride()):
        dims = sorted(((st, n) for n, st in zip(a.shape, a.stride()) if n > 1), reverse=True)

        def reachable(delta, i):      # delta = sum of d_i * stride_i with |d_i| < n_i over the dimensions from i on?
            if i == len(dims):
                return delta == 0
            st, n = dims[i]
            q = delta // st
            return any(abs(c) < n and reachable(abs(delta - c * st), i + 1) for c in (q, q + 1))

        return reachable(abs(a.storage_offset() - b.storage_offset()), 0)

    def span(t):
        lo = t.storage_offset() + sum((n - 1) * st for n, st in zip(t.shape, t.stride()) if st < 0)
        return lo, lo + 1 + sum((n - 1) * abs(st) for n, st in zip(t.shape, t.stride()))

    (a0, a1), (b0, b1) = span(a), span(b)
    return a0 < b1 and b0 < a1


@_on_device
def ballistics(u, z_alpha, coefficients=False, schedule=None, flags=None):
    """Ballistics.forward (core/envelope.py:84-101) on (R, L) rows: the float32 sequential recursion, bit for bit, whichever
    schedule produces it.  ``coefficients``: ``z_alpha`` holds (at, rt) themselves instead of their logits.
    ``flags``: a list that receives the per-row int32 flags of the chunked schedule (1 = the row was walked whole by the
    last launch: its coefficient is too slow for a chunk, or a chunk boundary failed the bit check) -- diagnostics."""
    schedule = BALLISTICS_SCHEDULE if schedule is None else schedule
    if schedule not in ("chunks", "rows"):
        raise ValueError(f"ballistics: unknown schedule {schedule!r}")
    _require_gpu(u, z_alpha)
    u, z_alpha = u.contiguous(), z_alpha.contiguous()
    R, L = u.shape
    if z_alpha.shape != (R, 2):
        raise ValueError(f"z_alpha must be ({R}, 2), got {tuple(z_alpha.shape)}")
    y = torch.empty_like(u)
    ws = None
    if schedule == "chunks":
        # zeroed: the single-pass schedules (short rows, very many rows) never write the flags this buffer is returned as
        ws = torch.zeros(lib().gfx_ballistics_ws_bytes(R), dtype=torch.uint8, device=u.device)
    with _timed("ballistics_walk_kernel", 8 * R * L):
        check(lib().gfx_ballistics_ws_f32(_ptr(u), _ptr(z_alpha), int(coefficients), _ptr(y), R, L, _ptr(ws),
                                          0 if ws is None else ws.numel(), _stream()), "gfx_ballistics_ws_f32")
    if flags is not None and ws is not None:
        flags.append(ws.view(torch.int32))
    return y


@_on_device
def dynamics_ballistics(x, log_threshold, log_ratio, log_knee, z_alpha, knee, gate, out=None, param_rows=None, schedule=None):
    """Compressor / NoiseGate with the ballistics energy smoother and no gain smoother in one pass over ``x`` ((R, C, L) or
    a strided (B, n, C, L) view): gfx_dynamics_ballistics_f32.  ``z_alpha``: (param_rows, 2)."""
    schedule = BALLISTICS_SCHEDULE if schedule is None else schedule
    _require_gpu(x, out, z_alpha)
    xmap, R, C, L = rowmap(x)
    P = R if param_rows is None else param_rows
    z_alpha = z_alpha.contiguous()
    if z_alpha.shape != (P, 2):
        raise ValueError(f"z_alpha must be ({P}, 2), got {tuple(z_alpha.shape)}")
    if out is None:
        out = torch.empty((R, C, L), dtype=torch.float32, device=x.device)
    elif rowmap(out)[1:] != (R, C, L):
        raise ValueError(f"dynamics_ballistics: output {tuple(out.shape)} does not match input rows/channels/length {(R, C, L)}")
    elif _overlap(x, out):
        # the chunked schedule writes every row in its first pass and re-reads x for the rows it has to walk again
        raise ValueError("dynamics_ballistics: out must not share memory with x (the kernel makes more than one pass over x)")
    ws = None
    if schedule == "chunks":
        ws = torch.empty(lib().gfx_ballistics_ws_bytes(R), dtype=torch.uint8, device=x.device)
    pin = _Pin()
    with _timed("ballistics_walk_kernel", 8 * R * C * L):
        check(lib().gfx_dynamics_ballistics_f32(_ptr(x), xmap, _ptr(out), rowmap(out)[0], pin(_rowvec(log_threshold, P)),
                                                pin(_rowvec(log_ratio, P)), pin(_rowvec(log_knee, P)), _ptr(z_alpha), P, R, C, L,
                                                KNEES[knee], int(gate), _ptr(ws), 0 if ws is None else ws.numel(), _stream()),
              "gfx_dynamics_ballistics_f32")
    return out


@_on_device
def ballistics_energy(x, z_alpha, coefficients=False, schedule=None):
    """ballistics(mean_c x^2) in one pass over x ((R, C, L) or a strided (B, n, C, L) view) -> (R, L): the envelope of
    Compressor / NoiseGate with energy_smoother="ballistics" (dynamics.py:390, core/envelope.py:84-101)."""
    schedule = BALLISTICS_SCHEDULE if schedule is None else schedule
    _require_gpu(x, z_alpha)
    xmap, R, C, L = rowmap(x)
    z_alpha = z_alpha.contiguous()
    if z_alpha.shape != (R, 2):
        raise ValueError(f"z_alpha must be ({R}, 2), got {tuple(z_alpha.shape)}")
    env = torch.empty((R, L), dtype=torch.float32, device=x.device)
    ws = None
    if schedule == "chunks":
        ws = torch.empty(lib().gfx_ballistics_ws_bytes(R), dtype=torch.uint8, device=x.device)
    with _timed("ballistics_walk_kernel", 4 * R * (C + 1) * L):
        check(lib().gfx_ballistics_energy_f32(_ptr(x), xmap, C, _ptr(z_alpha), int(coefficients), _ptr(env), R, L, _ptr(ws),
                                              0 if ws is None else ws.numel(), _stream()), "gfx_ballistics_energy_f32")
    return env


@_on_device
def ballistics_bwd(x, y, g, z_alpha, schedule="chunks"):
    """Adjoint of :func:`ballistics`: -> (dL/dx (R,L), dL/dz_alpha (R,2)).  ``schedule``: "chunks" (rows cut into chunks
    with a 2048-sample warm-up, gfx_ballistics_bwd_ws_f32) or "rows" (every row walked whole by one lane)."""
    _require_gpu(x, y, g, z_alpha)
    x, y, g, z_alpha = x.contiguous(), y.contiguous(), g.contiguous(), z_alpha.contiguous()
    R, L = x.shape
    _expect(y, (R, L), "ballistics_bwd: y")
    _expect(g, (R, L), "ballistics_bwd: g")
    _expect(z_alpha, (R, 2), "ballistics_bwd: z_alpha")
    gx, gz = torch.empty_like(x), torch.empty((R, 2), dtype=torch.float32, device=x.device)
    if schedule == "rows":
        check(lib().gfx_ballistics_bwd_f32(_ptr(x), _ptr(y), _ptr(g), _ptr(z_alpha), _ptr(gx), _ptr(gz), R, L, _stream()),
              "gfx_ballistics_bwd_f32")
        return gx, gz
    ws = torch.empty(max(int(lib().gfx_ballistics_bwd_ws_bytes(R, L)), 4), dtype=torch.uint8, device=x.device)
    check(lib().gfx_ballistics_bwd_ws_f32(_ptr(x), _ptr(y), _ptr(g), _ptr(z_alpha), _ptr(gx), _ptr(gz), R, L, _ptr(ws),
                                          ws.numel(), _stream()), "gfx_ballistics_bwd_ws_f32")
    return gx, gz


@_on_device
def dyn_gain(env, log_threshold, log_ratio, log_knee, knee, gate, log_out):
    _require_gpu(env)
    env = env.contiguous()
    R, L = env.shape
    g = torch.empty_like(env)
    pin = _Pin()
    check(
        lib().gfx_dyn_gain_f32(_ptr(env), _ptr(g), pin(_rowvec(log_threshold, R)), pin(_rowvec(log_ratio, R)),
                               pin(_rowvec(log_knee, R)), R, L, KNEES[knee], int(gate), int(log_out), _stream()),
        "gfx_dyn_gain_f32",
    )
    return g


@_on_device
def apply_gain(x, g, exp_gain=False, out=None):
    _require_gpu(x, g, out)
    xmap, R, C, L = rowmap(x)
    _expect(g, (R, L), "apply_gain: gain")
    g = g.contiguous()
    if out is None:
        out = torch.empty((R, C, L), dtype=torch.float32, device=x.device)
    elif rowmap(out)[1:] != (R, C, L):
        raise ValueError(f"apply_gain: output {tuple(out.shape)} does not match input rows/channels/length {(R, C, L)}")
    check(lib().gfx_apply_gain_f32(_ptr(x), xmap, _ptr(g), _ptr(out), rowmap(out)[0], R, C, L, int(exp_gain), _stream()), "gfx_apply_gain_f32")
    return out


@_on_device
def dyn_gain_apply(x, env, log_threshold, log_ratio, log_knee, knee, gate, out=None, param_rows=None):
    """y = exp(g(log(env + 1e-5)))[:, None, :] * x in one pass (dynamics.py:394-405): ``env`` (R, L) is the smoothed energy,
    ``x`` / ``out`` (R, C, L) tensors or strided (B, n, C, L) views."""
    _require_gpu(x, env, out)
    xmap, R, C, L = rowmap(x)
    _expect(env, (R, L), "dyn_gain_apply: envelope")
    env = env.contiguous()
    P = R if param_rows is None else param_rows
    if out is None:
        out = torch.empty((R, C, L), dtype=torch.float32, device=x.device)
    elif rowmap(out)[1:] != (R, C, L):
        raise ValueError(f"dyn_gain_apply: output {tuple(out.shape)} does not match input rows/channels/length {(R, C, L)}")
    pin = _Pin()
    with _timed("dyn_gain_apply_kernel", 4 * R * L * (2 * C + 1)):
        check(lib().gfx_dyn_gain_apply_f32(_ptr(x), xmap, _ptr(env), _ptr(out), rowmap(out)[0], pin(_rowvec(log_threshold, P)),
                                           pin(_rowvec(log_ratio, P)), pin(_rowvec(log_knee, P)), P, R, C, L, KNEES[knee],
                                           int(gate), _stream()), "gfx_dyn_gain_apply_f32")
    return out


@_on_device
def stereo_gain(x, log_gain, out=None, mix=None):
    """``mix``: as in :func:`dynamics_fused` -- the routing sum behind a strided (B, n, C, L) stage, produced by the same
    kernel (gfx_stereo_gain_mix_f32); sets ``mix["done"]`` when it was."""
    _require_gpu(x, log_gain, out)
    xmap, R, C, L = rowmap(x)
    _expect(log_gain, (R, 2), "stereo_gain: log_gain")
    if out is None:
        out = torch.empty((R, 2, L), dtype=torch.float32, device=x.device)
    elif rowmap(out)[1:] != (R, 2, L):
        raise ValueError(f"stereo_gain: output {tuple(out.shape)} does not match {(R, 2, L)}")
    pin = _Pin()
    if mix is not None and MIX_FUSION and x.ndim == 4 and out.ndim == 4:
        mo, sched = mix["out"], mix["sched"]
        n = x.shape[1]
        if mo.shape == (x.shape[0], mo.shape[1], 2, L) and mo.stride(-1) == 1 and sched.numel() == n:
            ex = mix.get("extras")
            n_ex = 0 if ex is None else ex.shape[0]
            n_pre = mix.get("n_pre", 0)
            rc = lib().gfx_stereo_gain_mix_f32(_ptr(x), xmap, pin(log_gain), _ptr(out), rowmap(out)[0], R, C, L, _ptr(sched), n,
                                               mix["n_acc"], _ptr(mo), mo.stride(0), mo.stride(1), mo.stride(2), _ptr(ex),
                                               n_pre, n_ex - n_pre, _stream())
            if rc == 0:
                mix["done"] = True
                return out
            if rc != -1:
                check(rc, "gfx_stereo_gain_mix_f32")
    check(lib().gfx_stereo_gain_f32(_ptr(x), xmap, pin(log_gain), _ptr(out), rowmap(out)[0], R, C, L, _stream()), "gfx_stereo_gain_f32")
    return out


@_on_device
def biquad_cascade(x, Bs, As, ssm_quirk=False, out=None):
    """Exact time-domain cascade of the K biquads in Bs/As (R,Cf,K,3) over x (R,C,L) or a (B,n,C,L) view."""
    _require_gpu(x, Bs, As, out)
    xmap, R, Cin, L = rowmap(x)
    Rb, Cf, K, three = Bs.shape
    if Rb != R or three != 3 or As.shape != Bs.shape:
        raise ValueError(f"coefficients {tuple(Bs.shape)} / {tuple(As.shape)} do not match {R} rows")
    Cout = max(Cin, Cf)
    if out is None:
        out = torch.empty((R, Cout, L), dtype=torch.float32, device=x.device)
    ymap = rowmap(out)[0]
    Bs, As = Bs.contiguous(), As.contiguous()
    with _timed("biquad_cascade_kernel", 4 * R * (Cin + Cout) * L):
        check(lib().gfx_biquad_cascade_f32(_ptr(x), xmap, _ptr(out), ymap, _ptr(Bs), _ptr(As), R, Cin, Cf, K, L,
                                           int(ssm_quirk), _stream()), "gfx_biquad_cascade_f32")
    return out


@_on_device
def noise_shaping_ir(noise, log_decay, log_gain, log_fade_in, z_fade_in_gain, ir_len, min_decay, max_decay):
    """noise (C,K,>=ir_len) view with unit last stride; parameters (R,C,K) -> ir (R,C,ir_len), un-normalised."""
    _require_gpu(noise, log_decay, log_gain, log_fade_in, z_fade_in_gain)
    R, C, K = log_decay.shape
    if noise.shape[:2] != (C, K) or noise.stride(-1) != 1 or noise.stride(1) != noise.stride(0) // K:
        raise ValueError("noise must be a (C, K, T) view of a contiguous band-split noise buffer")
    pin = _Pin()
    ir = torch.empty((R, C, ir_len), dtype=torch.float32, device=log_decay.device)
    check(
        lib().gfx_noise_shaping_ir_f32(_ptr(noise), noise.stride(1), pin(log_decay), pin(log_gain), pin(log_fade_in),
                                       pin(z_fade_in_gain), _ptr(ir), R, C, K, ir_len, float(min_decay), float(max_decay), _stream()),
        "gfx_noise_shaping_ir_f32",
    )
    return ir


# ----------------------------------------------------------------------------------------- waveshapers
WS_TANH, WS_PIECEWISE, WS_POWER, WS_CHEBYSHEV = 0, 1, 2, 3


@_on_device
def row_mean(x):
    """Mean over time of every row-channel: (R,C,L) or (B,n,C,L) view -> (R*C,)."""
    _require_gpu(x)
    xmap, R, C, L = rowmap(x)
    mean = torch.empty(R * C, dtype=torch.float32, device=x.device)
    check(lib().gfx_row_mean_f32(_ptr(x), xmap, _ptr(mean), R, C, L, _stream()), "gfx_row_mean_f32")
    return mean


@_on_device
def waveshaper(x, mode, log_pre_gain=None, log_post_gain=None, p0=None, p1=None, use_tanh=False,
               inverse_post_gain=False, remove_dc=False, out=None):
    """Memoryless distortion of every row (see gfx_waveshaper_f32 in the header for the modes)."""
    _require_gpu(x, log_pre_gain, log_post_gain, p0, p1, out)
    xmap, R, C, L = rowmap(x)
    if out is None:
        out = torch.empty((R, C, L), dtype=torch.float32, device=x.device)
    ymap = rowmap(out)[0]
    c = lambda t: None if t is None else t.contiguous().view(R, -1)  # noqa: E731
    log_pre_gain, log_post_gain, p0, p1 = c(log_pre_gain), c(log_post_gain), c(p0), c(p1)
    K = p0.shape[1] if (p0 is not None and mode in (WS_POWER, WS_CHEBYSHEV)) else 0
    dc = row_mean(x) if remove_dc else None
    with _timed("waveshaper_kernel", 8 * R * C * L):
        check(
            lib().gfx_waveshaper_f32(_ptr(x), xmap, _ptr(out), ymap, R, C, L, mode, int(use_tanh), int(inverse_post_gain),
                                     _ptr(log_pre_gain), _ptr(log_post_gain), _ptr(p0), _ptr(p1), K, _ptr(dc), _stream()),
            "gfx_waveshaper_f32",
        )
    return out


# ----------------------------------------------------------------------------------------- reverb IR
@_on_device
def istft_basis(window):
    _require_gpu(window)
    n_fft = window.numel()
    basis = torch.empty(lib().gfx_istft_basis_bytes(n_fft) // 4, dtype=torch.float32, device=window.device)
    pin = _Pin()
    check(lib().gfx_istft_basis_f32(pin(window), _ptr(basis), n_fft, _stream()), "gfx_istft_basis_f32")
    return basis


@_on_device
def stft(x, window, hop):
    """torch.stft(x, n_fft, hop, window, center=True, pad_mode="reflect", return_complex=True) for rows x (rows, T) on the
    direct-sum kernel (gfx_stft_f32): -> (rows, n_fft // 2 + 1, 1 + T // hop) complex64."""
    _require_gpu(x, window)
    x = x.contiguous()
    rows, T = x.shape
    n_fft = window.numel()
    frames = 1 + T // hop
    out = torch.empty((rows, n_fft // 2 + 1, frames, 2), dtype=torch.float32, device=x.device)
    pin = _Pin()
    for i in range(0, rows, 65535):     # rows ride on a grid dimension
        n = min(65535, rows - i)
        check(lib().gfx_stft_f32(_ptr(x[i:]), pin(window), _ptr(out[i:]), n, T, n_fft, hop, _stream()), "gfx_stft_f32")
    return torch.view_as_complex(out)


ISTFT_SCHEDULES = {"auto": 0, "gemm": 1, "fft": 2}   # GFX_ISTFT_* (include/grafx_amd.h)


@_on_device
def stft_reverb_ir(noise_stft, init_lm, delta_lm, gain_env, window, basis, ir_len, hop, ms_to_lr, schedule="auto"):
    """-> (ir (R,2,ir_len) un-normalised, row_gain (R) = 1/sqrt(mean_c sum_t ir^2 + 1e-12)).  `schedule`: how the frames
    are transformed -- "gemm" (matrix cores, any even n_fft), "fft" (n_fft 384 / hop 192 only), "auto"."""
    _require_gpu(init_lm, delta_lm, gain_env, window, basis)
    R = init_lm.shape[0]
    n_fft = window.numel()
    T = noise_stft.shape[-1]
    noise_rows = noise_stft.shape[0] if noise_stft.ndim == 4 else 1   # (R,2,K,T): fresh noise per row; else shared
    if noise_rows not in (1, R):
        raise ValueError(f"stft_reverb_ir: {noise_rows} noise spectra for {R} rows")
    nz = torch.view_as_real(noise_stft.contiguous())  # (…,2,K,T,2) float32 view of the complex64 buffer
    ir = torch.empty((R, 2, ir_len), dtype=torch.float32, device=init_lm.device)
    row_gain = torch.empty((R,), dtype=torch.float32, device=init_lm.device)
    sched = ISTFT_SCHEDULES[schedule]
    init_lm, delta_lm = init_lm.contiguous(), delta_lm.contiguous()
    gain_env = None if gain_env is None else gain_env.contiguous()
    ROWS = 32767                                        # rows per launch (the kernels' grids index rows in 16 bits)
    for r0 in range(0, R, ROWS):
        n = min(ROWS, R - r0)
        nbytes = lib().gfx_stft_reverb_workspace_bytes_sched(n, ir_len, n_fft, hop, T, sched)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=init_lm.device)
        pin = _Pin()
        check(
            lib().gfx_stft_reverb_ir_sched_f32(_ptr(nz[r0:r0 + n] if noise_rows != 1 else nz), noise_rows if noise_rows == 1 else n,
                                               pin(init_lm[r0:r0 + n]), pin(delta_lm[r0:r0 + n]),
                                               pin(None if gain_env is None else gain_env[r0:r0 + n]), pin(window), pin(basis),
                                               _ptr(ir[r0:r0 + n]), _ptr(row_gain[r0:r0 + n]), n, ir_len, n_fft, hop, T,
                                               int(ms_to_lr), _ptr(ws), nbytes, sched, _stream()),
            "gfx_stft_reverb_ir_sched_f32",
        )
    return ir, row_gain


# ----------------------------------------------------------------------------------------- routing
@_on_device
def gather_sum(buf, src_idx, seg_ptr, out):
    """out[b,j] = sum of buf[b, src_idx[e]] over e in [seg_ptr[j], seg_ptr[j+1]); buf/out are (B,V,C,L) views."""
    _require_gpu(buf, out)
    if buf.stride(-1) != 1 or out.stride(-1) != 1:
        raise ValueError("last dimension must be contiguous")
    B, _, C, L = buf.shape
    J = out.shape[1]
    if (out.shape[0], out.shape[2], out.shape[3]) != (B, C, L) or seg_ptr.numel() != J + 1:
        raise ValueError(f"gather_sum: output {tuple(out.shape)} / {seg_ptr.numel() - 1} segments do not match the "
                         f"buffer {tuple(buf.shape)}")
    with _timed("gather_sum_kernel", 4 * B * C * L * (src_idx.numel() + J)):
        check(
            lib().gfx_gather_sum_f32(_ptr(buf), buf.stride(0), buf.stride(1), buf.stride(2), _ptr(src_idx), _ptr(seg_ptr),
                                     _ptr(out), out.stride(0), out.stride(1), out.stride(2), B, J, C, L, _stream()),
            "gfx_gather_sum_f32",
        )
    return out


@_on_device
def gather_sum_fanout(buf, unique_src, dest_mask, out):
    """Fan-out form of :func:`gather_sum`: source rows read once, up to 32 destinations (bit mask per source)."""
    _require_gpu(buf, out)
    B, _, C, L = buf.shape
    J = out.shape[1]
    if (out.shape[0], out.shape[2], out.shape[3]) != (B, C, L) or dest_mask.numel() != unique_src.numel():
        raise ValueError(f"gather_sum_fanout: output {tuple(out.shape)} does not match the buffer {tuple(buf.shape)}")
    with _timed("gather_sum_kernel", 4 * B * C * L * (unique_src.numel() + J)):
        code = lib().gfx_gather_sum_fanout_f32(_ptr(buf), buf.stride(0), buf.stride(1), buf.stride(2), _ptr(unique_src),
                                               _ptr(dest_mask), unique_src.numel(), _ptr(out), out.stride(0),
                                               out.stride(1), out.stride(2), B, J, C, L, _stream())
    return code == 0
