"""Ballistics (core/envelope.py:84-101 -> torchcomp.compressor_core) on the HIP path: csrc/ballistics.hip.

The kernels promise the float32 SEQUENTIAL recursion bit for bit, whichever schedule produces it (rows cut into verified
chunks, or rows walked whole), so the comparisons here are exact equality against the oracle's float32 loop
(oracle.ballistics_coefficients: numpy, one rounding per product and per sum) with the SAME float32 coefficients handed
to both sides.  The third-party recursion itself is unpinned (oracle.ballistics docstring): every test here is
`provisional` by name, like every other ballistics test."""
import numpy as np
import pytest
import torch

import oracle
from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _coef(R, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(R, 2, generator=g) * (hi - lo) + lo).float()


def _run(u, coef, schedule):
    from grafx_amd import ops

    return ops.ballistics(u.cuda(), coef.cuda(), coefficients=True, schedule=schedule).cpu()


def _bits(t):
    return t.contiguous().view(torch.int32)


# (rows, length): whole tiles, ragged tails, L % 4 != 0 (element-wise loads), fewer rows than a wave, rows of several waves,
# rows that share a wave (few chunks per row), one row cut into 64 chunks
SHAPES = [(1, 1), (3, 64), (5, 1001), (2, 4096), (64, 4100), (65, 8192), (300, 16384), (7, 65536), (4097, 1024), (3, 131072)]


@pytest.mark.parametrize("R,L", SHAPES)
@pytest.mark.parametrize("schedule", ["chunks", "rows"])
def test_ballistics_is_the_float32_sequential_recursion_bit_for_bit(R, L, schedule):
    torch.manual_seed(R * 131 + L)
    u = torch.rand(R, L) * 2.0          # an energy-like input around the initial state y[-1] = 1
    coef = _coef(R, 0.02, 0.98, R + L)
    ref = oracle.ballistics_coefficients(u, coef[:, 0], coef[:, 1])
    y = _run(u, coef, schedule)
    assert torch.equal(_bits(y), _bits(ref)), f"{(y - ref).abs().max():.3e}"


@pytest.mark.parametrize("R,L", [(9, 32768), (130, 16384)])
def test_slow_coefficients_take_the_whole_row_walk_and_stay_exact(R, L):
    """Coefficients whose warm-up does not fit a chunk (the device decides per row): flagged, walked whole; and rows of
    both kinds in the same launch, in the same wave."""
    torch.manual_seed(R)
    u = torch.rand(R, L) * 3.0
    coef = _coef(R, 0.3, 0.9, 5)
    coef[::2] = _coef(R, 1e-4, 3e-3, 6)[::2]           # every other row slow
    coef[1, 0] = 1.0                                    # at = 1 exactly: y = x whenever it falls
    ref = oracle.ballistics_coefficients(u, coef[:, 0], coef[:, 1])
    for schedule in ("chunks", "rows"):
        y = _run(u, coef, schedule)
        assert torch.equal(_bits(y), _bits(ref)), schedule


@pytest.mark.parametrize("R,L", [(9, 131072), (300, 65536)])
def test_medium_coefficients_take_the_longer_chunks(R, L):
    """Coefficients of a few 1e-3: too slow for the first cut (2048 / 4096-sample chunks), fast enough for chunks eight
    times as long (the second launch, over the flagged rows); rows of the fast and of the very slow kind beside them."""
    torch.manual_seed(R + 1)
    u = torch.rand(R, L) * 2.0
    coef = _coef(R, 2e-3, 6e-3, 21)
    coef[1::3] = _coef(R, 0.2, 0.8, 22)[1::3]
    coef[2::7] = _coef(R, 5e-5, 2e-4, 23)[2::7]
    ref = oracle.ballistics_coefficients(u, coef[:, 0], coef[:, 1])
    y = _run(u, coef, "chunks")
    assert torch.equal(_bits(y), _bits(ref))


@pytest.mark.parametrize("lo,hi", [(0.3, 0.7), (0.01, 0.02), (2e-3, 4e-3)])
def test_verified_chunks_actually_carry_the_rows(lo, hi):
    """Exactness holds whichever path produces a row; SPEED needs the chunks to pass their bit check.  With the warm-up the
    kernel gives itself (34 / c steps: contraction to an ulp, then the merge of neighbouring floats) no row of 9216 may
    fall back to the whole-row walk at coefficients of 0.5 (first cut: 4096-sample chunks), 0.015 (still the first cut) and
    3e-3 (the second cut: 32768-sample chunks) -- one flagged row costs the launch the full 4.4 ms."""
    from grafx_amd import ops

    R, L = 9216, 131072
    torch.manual_seed(5)
    u = torch.rand(R, L, device="cuda") * 2
    coef = _coef(R, lo, hi, 3).cuda()
    flags = []
    ops.ballistics(u, coef, coefficients=True, schedule="chunks", flags=flags)
    assert int(flags[0].sum()) == 0, f"{int(flags[0].sum())} of {R} rows were walked whole"


def test_a_chunk_whose_warm_up_has_not_converged_is_caught_and_redone():
    """A 1e30 spike shortly before a chunk boundary: the true state is still ~1e22 where the next chunk starts, the
    warmed-up guess is ~1 -- the bit comparison of the two must flag the row, and the second launch walks it whole."""
    R, L = 4, 65536                       # 4 rows -> 64 chunks of 1024 samples
    torch.manual_seed(3)
    u = torch.rand(R, L)
    coef = torch.full((R, 2), 0.5)
    u[1, 3 * 1024 - 70] = 1e30            # 70 samples before chunk 3, whose warm-up covers only the last 64: not seen by it
    u[2, 17 * 1024 - 200] = 3e30
    ref = oracle.ballistics_coefficients(u, coef[:, 0], coef[:, 1])
    y = _run(u, coef, "chunks")
    assert torch.equal(_bits(y), _bits(ref))
    assert float(ref[1, 3 * 1024]) > 1e3  # (the spike really reaches across the boundary)


def test_negative_and_signed_inputs_nan_free():
    """The gain smoother feeds log-gains (negative) through the same recursion (dynamics.py:411-419)."""
    R, L = 33, 5000
    torch.manual_seed(9)
    u = torch.randn(R, L) * 4.0 - 3.0
    coef = _coef(R, 0.05, 0.95, 11)
    ref = oracle.ballistics_coefficients(u, coef[:, 0], coef[:, 1])
    for schedule in ("chunks", "rows"):
        assert torch.equal(_bits(_run(u, coef, schedule)), _bits(ref)), schedule


def test_logit_form_matches_the_oracle_with_its_own_sigmoid():
    """z_alpha -> sigmoid on the device (expf) vs torch's CPU sigmoid: an ulp of the coefficient, 1e-6 of the output."""
    from grafx_amd.processors import Ballistics

    R, L = 40, 20000
    torch.manual_seed(4)
    u, z = torch.rand(R, L) * 2, torch.randn(R, 2) * 2
    with torch.no_grad():
        y = Ballistics()(u.cuda(), z.cuda()).cpu()
    ref = oracle.ballistics(u, z)
    assert ((y - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("C", [1, 2])
@pytest.mark.parametrize("R,L", [(6, 4096), (70, 10000), (3, 65536)])
def test_energy_source_is_exact_too(R, L, C):
    """env = ballistics(mean_c x^2) in one pass over x: squares, channel sum and mean rounded as torch does
    (x.square().mean(-2), dynamics.py:390), then the same recursion -- bit-equal to the oracle on the float32 energy."""
    from grafx_amd import ops

    torch.manual_seed(R + L + C)
    x = torch.randn(R, C, L)
    coef = _coef(R, 0.05, 0.9, C)
    e = x.square().mean(-2)
    ref = oracle.ballistics_coefficients(e, coef[:, 0], coef[:, 1])
    for schedule in ("chunks", "rows"):
        y = ops.ballistics_energy(x.cuda(), coef.cuda(), coefficients=True, schedule=schedule).cpu()
        assert torch.equal(_bits(y), _bits(ref)), schedule


def test_energy_source_reads_a_strided_buffer_view_in_place():
    from grafx_amd import ops

    B, V, n, C, L = 3, 7, 4, 2, 8192
    torch.manual_seed(0)
    buf = torch.randn(B, V, C, L, device="cuda")
    view = buf[:, 2:2 + n]
    coef = _coef(B * n, 0.1, 0.9, 1)
    y = ops.ballistics_energy(view, coef.cuda(), coefficients=True).cpu()
    e = view.cpu().reshape(B * n, C, L).square().mean(-2)
    assert torch.equal(_bits(y), _bits(oracle.ballistics_coefficients(e, coef[:, 0], coef[:, 1])))


def test_chunked_and_whole_row_schedules_agree_at_the_console_size():
    """BASELINE configs[3]'s compressor rows (9216 x 131072): too long for the CPU loop in full, so the size-independent
    property -- both schedules give the same bits -- plus the oracle on a sample of rows."""
    R, L = 9216, 131072
    torch.manual_seed(1)
    u = torch.rand(R, L, device="cuda") * 2
    z = torch.randn(R, 2, device="cuda") * 0.1           # bench.py's parameter scale: coefficients ~ 0.5
    z[::97] = -7.0                                         # and a few long time constants (sigmoid(-7) ~ 9e-4)
    from grafx_amd import ops

    a = ops.ballistics(u, z, schedule="chunks")
    b = ops.ballistics(u, z, schedule="rows")
    assert torch.equal(_bits(a), _bits(b))
    rows = [0, 1, 97, 4607, 9215]
    coef = torch.sigmoid(z[rows].cpu())
    ours = ops.ballistics(u[rows].contiguous(), coef.cuda(), coefficients=True).cpu()
    ref = oracle.ballistics_coefficients(u[rows].cpu(), coef[:, 0], coef[:, 1])
    assert torch.equal(_bits(ours), _bits(ref))


@pytest.mark.parametrize("knee", ["hard", "quadratic", "exponential"])
@pytest.mark.parametrize("gate", [False, True])
def test_compressor_with_the_ballistics_smoother_matches_the_oracle(knee, gate):
    """Compressor / NoiseGate(energy_smoother="ballistics"): ballistics_energy + the fused gain stage (dyn_gain_apply)."""
    import grafx_amd.processors as P

    R, C, L = 12, 2, 12000
    torch.manual_seed(int(gate) * 10 + len(knee))
    cls, ocls = (P.NoiseGate, oracle.OracleNoiseGate) if gate else (P.Compressor, oracle.OracleCompressor)
    m = cls(energy_smoother="ballistics", knee=knee, flashfftconv=False).cuda()
    o = ocls(energy_smoother="ballistics", knee=knee)
    x = torch.randn(R, C, L) * 0.3
    p = {k: torch.randn(R, *((v,) if isinstance(v, int) else v)) for k, v in m.parameter_size().items()}
    with torch.no_grad():
        y = m(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
    ref = o(x, **p)
    ref64 = o(x.double(), **{k: v.double() for k, v in p.items()}).float()
    assert_parity(y, ref, ref64, 1e-5, f"{cls.__name__} ballistics {knee}")


@pytest.mark.parametrize("R,L", [(130, 40000), (64, 131072), (200, 20001)])
def test_chunked_adjoint_equals_the_whole_row_adjoint(R, L):
    """gfx_ballistics_bwd_ws_f32: the adjoint recursion lambda[n] = g[n] + (1 - c[n+1]) lambda[n+1] is linear and a
    contraction, so chunks that start 2048 samples later with a zero carry reproduce the whole-row walk to (1 - c)^2048;
    groups of 64 rows with a coefficient below 0.0103 are walked whole.  Rows of both kinds, ragged last group, L % 4 != 0;
    the coefficient gradients are sums over the row (per-chunk partials, added in chunk order)."""
    from grafx_amd import ops

    torch.manual_seed(R + L)
    x = torch.rand(R, L, device="cuda") * 2
    z = torch.randn(R, 2, device="cuda") * 1.5
    if R > 70:
        z[70:] = torch.randn(R - 70, 2, device="cuda") * 0.5 - 6.0  # the second group onwards: coefficients ~ 2.5e-3 (whole rows)
    z[3, 0] = -4.0                                                    # one slow row in the first group (c = 0.018: still chunked)
    y = ops.ballistics(x, z)
    g = torch.randn(R, L, device="cuda")
    gx_c, gz_c = ops.ballistics_bwd(x, y, g, z, schedule="chunks")
    gx_r, gz_r = ops.ballistics_bwd(x, y, g, z, schedule="rows")
    assert (gx_c - gx_r).abs().max() <= 1e-6 * gx_r.abs().max(), float((gx_c - gx_r).abs().max() / gx_r.abs().max())
    assert (gz_c - gz_r).abs().max() <= 2e-5 * gz_r.abs().max(), float((gz_c - gz_r).abs().max() / gz_r.abs().max())
    again = ops.ballistics_bwd(x, y, g, z, schedule="chunks")
    assert torch.equal(again[0], gx_c) and torch.equal(again[1], gz_c)      # fixed summation order: the same bits


def test_chunked_adjoint_at_the_console_size_against_float64_on_sample_rows():
    """9216 x 131072 (BASELINE configs[3]'s compressor rows): the chunked adjoint on the device against the float64 adjoint
    recursion on the CPU for a few rows."""
    from grafx_amd import ops

    R, L = 9216, 131072
    torch.manual_seed(2)
    x = torch.rand(R, L, device="cuda") * 2
    z = torch.randn(R, 2, device="cuda") * 0.1
    y = ops.ballistics(x, z)
    g = torch.randn(R, L, device="cuda")
    gx, gz = ops.ballistics_bwd(x, y, g, z)
    for r in (0, 4607, 9215):
        xs, ys, gs = x[r].double().cpu().numpy(), y[r].double().cpu().numpy(), g[r].double().cpu().numpy()
        at, rt = torch.sigmoid(z[r].double()).cpu().tolist()
        lam, carry, sa, sr = np.zeros(L), 0.0, 0.0, 0.0
        for n in range(L - 1, -1, -1):
            yp = ys[n - 1] if n > 0 else 1.0
            attack = xs[n] < yp
            c = at if attack else rt
            l_ = gs[n] + carry
            lam[n] = c * l_
            d = l_ * (xs[n] - yp)
            sa, sr = (sa + d, sr) if attack else (sa, sr + d)
            carry = (1.0 - c) * l_
        ref = torch.from_numpy(lam).float()
        assert (gx[r].cpu() - ref).abs().max() <= 2e-6 * ref.abs().max(), r
        want = torch.tensor([sa * at * (1 - at), sr * rt * (1 - rt)])
        assert ((gz[r].cpu().double() - want).abs() <= 1e-4 * want.abs().max()).all(), (r, gz[r].cpu(), want)


def test_in_place_output_is_refused_and_short_rows_return_clean_flags():
    """gfx_dynamics_ballistics_f32 makes more than one pass over x (rows that fail the bit check are walked again from the
    input), so an output that shares memory with the input is an error, not a silent wrong answer; two node ranges of one
    signal buffer -- interleaved in memory, never touching -- are fine.  And the diagnostics flags of a call that took a
    single-pass schedule (rows shorter than a chunk) are zeros, not uninitialised memory."""
    from grafx_amd import ops

    torch.manual_seed(0)
    R, C, L = 6, 2, 4096
    x = torch.randn(R, C, L, device="cuda")
    p = [torch.randn(R, 1, device="cuda") for _ in range(3)]
    za = torch.randn(R, 2, device="cuda")
    want = ops.dynamics_ballistics(x, *p, za, "quadratic", False)
    with pytest.raises(ValueError, match="share memory"):
        ops.dynamics_ballistics(x, *p, za, "quadratic", False, out=x)
    buf = torch.zeros(2, 8, C, L, device="cuda")
    buf[:, 1:4] = x.view(2, 3, C, L)
    ops.dynamics_ballistics(buf[:, 1:4], *p, za, "quadratic", False, out=buf[:, 4:7])
    assert torch.equal(buf[:, 4:7].reshape(R, C, L), want)
    with pytest.raises(ValueError, match="share memory"):
        ops.dynamics_ballistics(buf[:, 1:4], *p, za, "quadratic", False, out=buf[:, 3:6])
    flags = []
    ops.ballistics(torch.rand(3, 100, device="cuda"), torch.randn(3, 2, device="cuda"), flags=flags)
    assert int(flags[0].abs().sum()) == 0
