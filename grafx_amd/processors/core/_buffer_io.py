"""In-place signal-buffer I/O protocol between render_grafx and the HIP processors.

``render_into(x4, out4, **params)`` receives strided (B, n, C, L) views of the render loop's
signal buffer (input slice, destination slice) and per-row parameters flattened batch-major
(B*n, ...), exactly the rows ``forward`` would get.  Processors whose kernels address rows
through a gfx_rowmap_t read and write the buffer directly; the default falls back to
``forward`` on a flattened copy.

``prepare(**params[, _shared_rows]) -> Prepared | None`` (optional): the parameter-only part of ``render_into``
(filter design, impulse-response synthesis, spectra).  The render loop runs it ahead of time on a side stream,
under the signal kernels of the earlier stages, and hands the result back as ``render_into(..., _prepared=...)``."""


class Prepared:
    """What ``prepare`` returns: device tensors (so the render can order their lifetime across streams) + scalars."""

    def __init__(self, *tensors, **scalars):
        self.tensors = tensors
        self.__dict__.update(scalars)


class BufferIO:
    def render_into(self, x4, out4, **params):
        y = self.forward(x4.reshape(-1, *x4.shape[2:]), **params)
        out4.copy_(y.view(out4.shape))
        return out4


def shared_reps(x, shared_rows):
    """How many times `shared_rows` parameter rows repeat over the rows of x ((R,C,L) or a (B,n,C,L) view)."""
    rows = x.shape[0] * x.shape[1] if x.ndim == 4 else x.shape[0]
    if rows % shared_rows != 0:
        raise ValueError(f"{rows} signal rows cannot share {shared_rows} parameter rows")
    return rows // shared_rows


def expand_shared(t, reps):
    """(n, ...) per-node parameter -> (reps*n, ...) batch-major rows (what upstream's expand + flatten produces)."""
    return None if t is None else t.repeat(reps, *([1] * (t.ndim - 1)))
