"""python tools/czt_chunk_ab.py: the odd-length aliasing (two chirp-z transforms per row) on the compat console's shapes with
different chunk sizes (rows per launch chain; the workspace is 2 MB per row at P = 135 071 in float, 4 MB in double)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from grafx_amd import ops

for P, rows, precise in [(135071, 4608, False), (147455, 2304, True)]:
    z = torch.randn(rows, P, device="cuda")
    for chunk, cap in [(256, 1 << 30), (512, 2 << 30), (1024, 4 << 30), (2304, 16 << 30), (4608, 16 << 30)]:
        ops.ALIAS_WS_CAP = cap
        ops.odd_alias(z, rows_per_chunk=chunk, precise=precise)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            ops.odd_alias(z, rows_per_chunk=chunk, precise=precise)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        print(f"P={P} rows={rows} precise={precise} chunk={chunk:5d}: {dt * 1e3:8.2f} ms", flush=True)
