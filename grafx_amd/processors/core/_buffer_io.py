"""In-place signal-buffer I/O protocol between render_grafx and the HIP processors.

``render_into(x4, out4, **params)`` receives strided (B, n, C, L) views of the render loop's
signal buffer (input slice, destination slice) and per-row parameters flattened batch-major
(B*n, ...), exactly the rows ``forward`` would get.  Processors whose kernels address rows
through a gfx_rowmap_t read and write the buffer directly; the default falls back to
``forward`` on a flattened copy."""


class BufferIO:
    def render_into(self, x4, out4, **params):
        y = self.forward(x4.reshape(-1, *x4.shape[2:]), **params)
        out4.copy_(y.view(out4.shape))
        return out4
