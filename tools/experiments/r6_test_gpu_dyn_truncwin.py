"""The compressor's smoother with a LIVE truncation term on the tile grid (csrc/dynamics.hip, "truncated-window" rows, round 6).

TruncatedOnePoleIIRFilter (reference core/envelope.py:34-60) is an N-tap FIR, h[k] = (1 - a) a^k for k < N.  When a^N is not
negligible (a > 0.9983 at 16383 taps; the clamp a = 1 - 1e-5 keeps 0.85 of its first tap at the last) the smoother is NOT a
one-pole recursion with a forgotten tail: through round 5 such rows left the tile grid for one workgroup per row, and a
fused routing sum read them back.  Now the state entering tile j is the window sum_{k<N} a^k e[s-1-k] -- the aggregates of the
q = N // 512 tiles before it plus the last N % 512 samples of tile j-q-1 (a second granule per tile) -- and the tile scans
e[n] - a^N e[n-N].  Checked: against the row kernel on the same rows (ops.dynamics_fused(schedule="rows")), against the
oracle with the float64 tie-breaker, tap counts with and without a partial tile, rows of all kinds in one launch, the fused
routing sum, mono / stereo, rows shorter than the window, repeated launches on a recycled workspace."""
import pytest
import torch

import oracle
from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _rows(R, C, L, seed):
    torch.manual_seed(seed)
    x = torch.randn(R, C, L, device="cuda") * torch.linspace(0.3, 1.0, R, device="cuda")[:, None, None]
    x[:, :, L // 3 : L // 3 + 2500] *= 2.0          # a burst the window has to carry and then DROP after N samples
    x[1, :, : L // 4] = 0                            # a silent start
    return x


@pytest.mark.parametrize("N", [16383, 4001, 1000, 700, 512, 1024, 300])
@pytest.mark.parametrize("C", [1, 2])
def test_truncated_window_tiles_equal_the_row_kernel(N, C):
    """Rows at and near the clamp through the tiles and through the row kernel: two orders of the same N-term sums."""
    from grafx_amd import ops

    R, L = 24, 65536 + 512 * 3
    x = _rows(R, C, L, N)
    # poles from just alive at this N to the clamp, a few fast and look-back ones mixed in (all three tile kinds + none on the row kernel)
    z = torch.cat([torch.linspace(5.0, 14.0, R - 4), torch.tensor([0.0, 3.0, -1.0, 20.0])]).to("cuda")[:, None]
    lt, lr, lk = (torch.randn(R, 1, device="cuda") for _ in range(3))
    got = ops.dynamics_fused(x, lt - 1, lr, lk, z, smoother=1, iir_len=N, knee="quadratic", gate=False)
    want = ops.dynamics_fused(x, lt - 1, lr, lk, z, smoother=1, iir_len=N, knee="quadratic", gate=False, schedule="rows")
    for r in range(R):
        err = (got[r] - want[r]).abs().max() / want[r].abs().max()
        assert float(err) <= 5e-6, f"row {r} z={float(z[r]):.2f}: {float(err):.2e}"
    again = ops.dynamics_fused(x, lt - 1, lr, lk, z, smoother=1, iir_len=N, knee="quadratic", gate=False)
    assert torch.equal(got, again)                   # recycled workspace: no granule of the previous launch is seen


@pytest.mark.parametrize("L,iir_len", [(131072, 16383), (20000, 16383), (40000, 4001), (3000, 4001)])
def test_compressor_at_the_clamp_matches_the_oracle(L, iir_len):
    from grafx_amd.processors import Compressor

    z = torch.tensor([[12.0], [20.0], [8.0], [7.0], [6.6], [9.5], [0.0], [5.0]])
    R = z.shape[0]
    torch.manual_seed(L + iir_len)
    x = torch.randn(R, 2, L) * torch.linspace(0.5, 1.0, R)[:, None, None]
    x[:, :, L // 2 : L // 2 + L // 40] *= 1.5
    g = torch.Generator().manual_seed(3)
    p = {"log_threshold": torch.randn(R, 1, generator=g) - 1, "log_ratio": torch.randn(R, 1, generator=g),
         "log_knee": torch.randn(R, 1, generator=g), "z_alpha_pre": z}
    m = Compressor(energy_smoother="iir", iir_len=iir_len, flashfftconv=False).cuda()
    with torch.no_grad():
        y = m(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
    o = oracle.OracleCompressor(iir_len=iir_len)
    ref = o(x, **p)
    ref64 = o(x.double(), **{k: v.double() for k, v in p.items()}).float()
    assert_parity(y, ref, ref64, 1e-5, f"compressor at the clamp L={L} iir_len={iir_len}")


@pytest.mark.parametrize("knee,gate", [("quadratic", False), ("hard", True), ("exponential", False)])
def test_truncated_window_rows_through_the_fused_routing_sum(knee, gate):
    """The console's strip stage with every smoother at the clamp: the rows and the bus sums from the tile walk equal the
    row kernel's rows and their gather-sum (bit for bit for the sums GIVEN the rows: same order of additions)."""
    from grafx_amd import ops

    B, n, C, L = 3, 8, 2, 32768
    torch.manual_seed(5)
    buf = torch.zeros(B, 2 * n + 3, C, L, device="cuda")
    buf[:, :n] = torch.randn(B, n, C, L, device="cuda") * 0.5
    z = torch.tensor([12.0, 9.0, 7.5, 12.0, 3.0, 0.0, 12.0, 6.0], device="cuda")[:, None]
    lt, lr, lk = (torch.randn(n, 1, device="cuda") for _ in range(3))
    dests = [[0, 1, 2, 3], [4, 5, 6, 7], list(range(8))]
    codes, n_acc, pre, post = ops.mix_schedule(dests, n)
    assert not pre and not post
    mix = {"sched": torch.tensor(codes, dtype=torch.long, device="cuda"), "n_acc": n_acc, "out": buf[:, 2 * n :]}
    ops.dynamics_fused(buf[:, :n], lt - 1, lr, lk, z, smoother=1, iir_len=16383, knee=knee, gate=gate, out=buf[:, n : 2 * n],
                       param_rows=n, mix=mix)
    assert mix.get("done")
    rows = ops.dynamics_fused(buf[:, :n].reshape(B * n, C, L).contiguous(), lt.repeat(B, 1) - 1, lr.repeat(B, 1), lk.repeat(B, 1),
                              z.repeat(B, 1), smoother=1, iir_len=16383, knee=knee, gate=gate, schedule="rows").view(B, n, C, L)
    assert (buf[:, n : 2 * n] - rows).abs().max() <= 5e-6 * rows.abs().max()
    for d, src in enumerate(dests):
        want = torch.zeros(B, C, L, device="cuda")
        for j in src:                                     # increasing order, from 0.0f: the kernels' order
            want = want + buf[:, n + j]
        assert torch.equal(buf[:, 2 * n + d], want), d


def test_a_kept_scan_sends_truncated_rows_back_to_the_row_kernel():
    """u1_out asks for the UN-truncated scan, which only the row kernel has for a row with a live truncation term: such
    calls keep round 5's path, and give the same output."""
    from grafx_amd import ops

    R, C, L, N = 6, 2, 32768, 4001
    x = _rows(R, C, L, 1)
    z = torch.tensor([[12.0], [8.0], [6.0], [0.0], [3.0], [10.0]], device="cuda")
    lt, lr, lk = (torch.randn(R, 1, device="cuda") for _ in range(3))
    u1 = torch.empty(R, L, device="cuda")
    a = ops.dynamics_fused(x, lt, lr, lk, z, smoother=1, iir_len=N, knee="quadratic", gate=False, u1_out=u1)
    b = ops.dynamics_fused(x, lt, lr, lk, z, smoother=1, iir_len=N, knee="quadratic", gate=False)
    assert (a - b).abs().max() <= 5e-6 * a.abs().max()
    assert torch.isfinite(u1).all()
