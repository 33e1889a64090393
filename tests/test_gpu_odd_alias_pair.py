"""Two real rows per complex chirp-z transform (csrc/czt_pair.hip) against torch.fft in float64 and against the one-row
transforms of czt.hip (reference: core/convolution.py:119-134, y = irfft_{P-1}(rfft_P(z))).

ops.odd_alias takes this form by default for calls of two rows or more (P <= 8 388 607), so tests/test_gpu_odd_alias.py and
every compat-length processor test run on it too; here: every tile count of the pair form, odd row counts (the last row
alone in its transform), rows of very different size in one pair (each row is held to ITS OWN peak: the second row of a
pair goes through scaled to the first one's binade), slices, strided in-place output, the double-precision form, and the
switch back to one transform per row."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _want(z):
    return torch.fft.irfft(torch.fft.rfft(z.double()))


def _one_row_form(fn):
    from grafx_amd import ops

    old = ops.ALIAS_PAIRS
    ops.ALIAS_PAIRS = False
    try:
        return fn()
    finally:
        ops.ALIAS_PAIRS = old


# P just below / above each boundary 2P - 1 = C x 8192 of the pair form's tile counts (C = 1 .. 63 with factors <= 7)
PAIR_SIZES = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16, 18, 20, 21, 24, 25, 27, 28, 30, 32, 35, 36, 40, 42, 45, 48, 49, 50,
              54, 56, 60, 63]


@pytest.mark.parametrize("C", PAIR_SIZES)
def test_every_tile_count_of_the_pair_form(C):
    from grafx_amd import ops
    from grafx_amd._lib import lib

    P = (C * 8192 + 1) // 2                     # the largest P with 2P - 1 <= C x 8192 ...
    P -= 1 - (P & 1)                            # ... that is odd
    assert lib().gfx_odd_alias_pair_workspace_bytes(2, P) == C * 8192 * 8 + 256, (C, P)    # (+ the rows' max |z| words)
    torch.manual_seed(C)
    z = torch.randn(5, P, device="cuda")        # an odd row count: two pairs and a single
    got = ops.odd_alias(z)
    want = _want(z)
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= 3e-6, f"C={C} P={P}: {err:.2e}"
    single = _one_row_form(lambda: ops.odd_alias(z))
    assert (got - single).abs().max() <= 3e-6 * single.abs().max()


@pytest.mark.parametrize("P", [3, 5, 7, 101, 4001, 8191, 8193, 135071, 147455, 191071, 258047])
def test_pairs_match_float64_fft_and_slices_are_bit_equal(P):
    from grafx_amd import ops

    torch.manual_seed(P)
    z = torch.randn(3, 2, P, device="cuda")
    got = ops.odd_alias(z)
    want = _want(z)
    assert got.shape == want.shape
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= 3e-6, f"P={P}: {err:.2e}"
    lo, n = P // 3, max(1, P // 5)
    assert torch.equal(ops.odd_alias(z, lo, n), got[..., lo : lo + n])


@pytest.mark.parametrize("P", [258049, 299999, 483999, 1000001, 2097153, 4200001])
def test_longer_rows_take_outer_radix_4_levels(P):
    """More than 63 tiles per pair: one, two and three outer radix-4 levels around czt.hip's column passes (BASELINE
    configs[1] / configs[2] with upstream's default tap counts: P = 483 999 and 299 999)."""
    from grafx_amd import ops
    from grafx_amd._lib import lib

    assert lib().gfx_odd_alias_pair_plan_bytes(P) > 0
    torch.manual_seed(P)
    z = torch.randn(3, P, device="cuda")
    got = ops.odd_alias(z)
    want = _want(z)
    err = (got.double() - want).abs().max() / want.abs().max()
    assert err <= (3e-6 if P < 700000 else 5e-6), f"P={P}: {err:.2e}"
    lo, n = P // 3, max(1, P // 5)
    assert torch.equal(ops.odd_alias(z, lo, n), got[..., lo : lo + n])
    if P < 500000:
        gotd = ops.odd_alias(z, precise=True)
        assert (gotd.double() - want).abs().max() <= 1.5e-7 * want.abs().max()


def test_beyond_the_pair_forms_reach_one_transform_per_row():
    from grafx_amd import ops
    from grafx_amd._lib import lib

    P = 8388609
    assert lib().gfx_odd_alias_pair_plan_bytes(P) == 0 and lib().gfx_odd_alias_pair_plan_bytes(P - 2) > 0
    assert lib().gfx_odd_alias_pair_plan_bytes(4000) == 0              # even lengths do not alias
    torch.manual_seed(0)
    z = torch.randn(2, P, device="cuda")
    got = ops.odd_alias(z)
    want = _want(z)
    assert (got.double() - want).abs().max() / want.abs().max() <= 5e-6


# 135 071: one column pass (35 tiles); 299 999 / 483 999: the fused outer level (BASELINE configs[2] / configs[1] at upstream's
# tap counts); 1 000 001: two outer levels on czt.hip's passes
@pytest.mark.parametrize("P", [4001, 135071, 299999, 483999, 1000001])
@pytest.mark.parametrize("ratio", [1e-3, 1e-6, 1e-12, 0.0, 1e3, 1e6])
def test_a_loud_and_a_quiet_row_in_one_pair(P, ratio):
    """One complex transform carries both rows of a pair, and its rounding error is eps times the larger component: the
    second row therefore goes in scaled by the power of two that brings it to the first row's binade and the scale is
    divided out of the result (csrc/czt_pair.hip, pair_scale).  Every row keeps an error relative to ITS OWN peak -- what the
    reference's independent rows give (core/convolution.py:119-134) -- whichever of the two is the quiet one, and a row
    that is all zero comes out all zero."""
    from grafx_amd import ops

    torch.manual_seed(1)
    z = torch.randn(4, P, device="cuda")
    z[1] *= ratio                # pair (0, 1): the second row quieter (or louder: ratio > 1) by `ratio`
    z[2] *= ratio                # pair (2, 3): the FIRST row is the odd one out
    got = ops.odd_alias(z)
    want = _want(z)
    tol = 3e-6 if P < 700000 else 5e-6
    for r in range(4):
        peak = want[r].abs().max()
        err = (got[r].double() - want[r]).abs().max()
        assert err <= tol * peak, f"row {r}: {float(err / peak.clamp_min(1e-300)):.2e} of its own peak"
        if ratio == 0.0 and r in (1, 2):
            assert float(got[r].abs().max()) == 0.0
    alone = _one_row_form(lambda: ops.odd_alias(z))
    for r in range(4):
        assert (alone[r].double() - want[r]).abs().max() <= tol * want[r].abs().max()


@pytest.mark.parametrize("precise", [False, True])
@pytest.mark.parametrize("P", [4001, 135071, 299999, 1000001])
def test_a_power_of_two_on_one_row_of_a_pair_changes_nothing_else(P, precise):
    """The scaling is exact: multiplying one row of a pair by 2^k multiplies that row of the result by 2^k, bit for bit,
    and leaves its partner's bits alone -- for either row of the pair (the first row's scale moves the SECOND row's
    internal representation, not its result)."""
    from grafx_amd import ops

    torch.manual_seed(P)
    z = torch.randn(2, P, device="cuda")
    base = ops.odd_alias(z, 7, 5000 if P > 5007 else None, precise=precise)
    for row in (0, 1):
        for k in (-40, -3, 1, 17):
            zz = z.clone()
            zz[row] *= 2.0 ** k
            got = ops.odd_alias(zz, 7, 5000 if P > 5007 else None, precise=precise)
            assert torch.equal(got[row], base[row] * 2.0 ** k), (row, k)
            assert torch.equal(got[1 - row], base[1 - row]), (row, k)


def test_rows_do_not_depend_on_their_neighbours_scale():
    """Batch invariance at the level the pair form allows: a row's result is the same bits whatever power of two its
    partner carries, and an all-zero partner is the same as no partner's contribution at all (exact zeros out)."""
    from grafx_amd import ops

    P = 20001
    torch.manual_seed(5)
    z = torch.randn(6, P, device="cuda")
    z[3] = 0
    z[4] = 0
    z[5] = 0
    got = ops.odd_alias(z)
    assert float(got[3:].abs().max()) == 0.0
    want = _want(z)
    for r in range(3):
        assert (got[r].double() - want[r]).abs().max() <= 3e-6 * want[r].abs().max()


@pytest.mark.parametrize("P", [101, 135071, 147455])
def test_precise_pairs_are_float64_accurate(P):
    """The double-precision form (the energy envelope's): fp32 rounding of the output only, even across a pair whose rows
    differ by 1e4 in size."""
    from grafx_amd import ops

    torch.manual_seed(P)
    z = torch.randn(7, P, device="cuda").abs()
    z[1] *= 1e-4
    z[4] *= 1e-9
    z[6] *= 1e-7
    got = ops.odd_alias(z, precise=True)
    want = _want(z)
    for r in range(z.shape[0]):
        assert (got[r].double() - want[r]).abs().max() <= 1.5e-7 * want[r].abs().max(), r
    lo, n = P // 4, P // 2
    assert torch.equal(ops.odd_alias(z, lo, n, precise=True), got[..., lo : lo + n])


@pytest.mark.parametrize("P,C", [(4001, 2), (135071, 2), (9001, 1), (9001, 3), (262145, 2), (1000001, 1)])
def test_pairs_write_strided_buffer_rows_in_place(P, C):
    """gfx_odd_alias_pair_rows_f32: rows 2r, 2r + 1 of the call -> row q / C, channel q % C of a strided (B, n, C, len) view
    (the render's signal buffer); odd C makes pairs straddle signal rows; chunks keep pairs whole.  P = 262 145: the fused
    one-outer-level kernels; 1 000 001: two outer levels on czt.hip's passes."""
    from grafx_amd import ops

    B, V, n, L = 2, 5, 3, P - 1 - 7
    torch.manual_seed(P + C)
    buf = torch.zeros(B, V, C, L, device="cuda")
    view = buf[:, 1 : 1 + n]
    z = torch.randn(B * n * C, P, device="cuda")
    ops.odd_alias(z, 3, L, out=view, rows_per_chunk=4)
    want = ops.odd_alias(z, 3, L).view(B, n, C, L)
    assert torch.equal(view, want)
    assert float(buf[:, 0].abs().max()) == 0.0 and float(buf[:, 1 + n :].abs().max()) == 0.0


def test_chunked_calls_give_the_same_bits():
    from grafx_amd import ops

    P = 20001
    torch.manual_seed(2)
    z = torch.randn(11, P, device="cuda")
    whole = ops.odd_alias(z)
    for per in (2, 3, 4, 10):
        assert torch.equal(ops.odd_alias(z, rows_per_chunk=per), whole), per


@pytest.mark.parametrize("P,rows,precise", [(135071, 64, False), (147455, 40, True), (20001, 420, False), (20001, 230, True)])
def test_a_call_of_many_rows_gives_the_bits_of_calls_of_two(P, rows, precise):
    """More tiles than the device holds at once (grids of thousands of workgroups, every pair at its own workspace offset)
    against the same rows two at a time: the same bits, in the two-row and in the one-row form."""
    from grafx_amd import ops

    torch.manual_seed(rows)
    z = torch.randn(rows, P, device="cuda")
    for pairs in (True, False):
        old = ops.ALIAS_PAIRS
        ops.ALIAS_PAIRS = pairs
        try:
            big = ops.odd_alias(z, 5, 1000, precise=precise)
            small = torch.cat([ops.odd_alias(z[i : i + 2], 5, 1000, precise=precise) for i in range(0, rows, 2)])
        finally:
            ops.ALIAS_PAIRS = old
        assert torch.equal(big, small), pairs
    want = torch.fft.irfft(torch.fft.rfft(z.double()))[:, 5:1005]
    assert (big.double() - want).abs().max() <= 3e-6 * want.abs().max()


@pytest.mark.parametrize("P", [258049, 299999, 483999])
def test_fused_outer_level_equals_the_separate_passes(P, monkeypatch):
    """One outer radix-4 level: czt_pair_lv_* (outer pass + column pass in one kernel, five sweeps) against the chain of
    separate passes (nine sweeps; GRAFX_CZT_FUSED_LEVEL=0 is read once per process, so the separate chain is reached through
    the one-row form here, which has no fused kernels): the same transform, another factorisation of the twiddles."""
    from grafx_amd import ops

    torch.manual_seed(P)
    z = torch.randn(5, P, device="cuda")
    got = ops.odd_alias(z, 100, 4096)
    single = _one_row_form(lambda: ops.odd_alias(z, 100, 4096))
    want = _want(z)[:, 100:4196]
    assert (got.double() - want).abs().max() <= 3e-6 * want.abs().max()
    assert (got - single).abs().max() <= 3e-6 * single.abs().max()


def test_the_convolution_kernel_leaves_the_rows_maxima_for_the_pair_scaling():
    """gfx_fftconv_rowmax_f32: the tile kernel of the full-length convolution keeps max |y| per output row-channel as a
    by-product of its stores, and the aliasing's pair form takes the words instead of reading z once more
    (gfx_odd_alias_pair_max_f32); the partitioned convolution (more than 8193 taps) does the same.  Rows at very different levels, a silent row; with and without the by-product: the same bits."""
    from grafx_amd import ops

    torch.manual_seed(0)
    R, C, L, N = 300, 2, 65536, 4000
    x = torch.randn(R, C, L, device="cuda") * torch.logspace(0, -6, R, device="cuda")[:, None, None]
    x[7] = 0
    h = torch.randn(R, 1, N, device="cuda") / N ** 0.5
    Hs = ops.fir_spectrum(h.reshape(R, N))
    rm = {}
    z = ops.fftconv(x, Hs, N, 1, Lout=L + N - 1, rowmax=rm)
    assert "words" in rm
    got = rm["words"].view(torch.float32).view(R, C)
    assert torch.equal(got, z.abs().amax(-1))                    # the maxima of exactly what was stored
    assert float(got[7].abs().max()) == 0.0
    y1 = ops.odd_alias(z, 0, L, rowmax=rm["words"])
    y0 = ops.odd_alias(z, 0, L)
    assert torch.equal(y1, y0)
    ref = _want(z)[..., :L]
    err = (y1.double() - ref).abs().amax(-1) / ref.abs().amax(-1).clamp_min(1e-300)
    assert float(err[torch.isfinite(err)].max()) <= 3e-6
    assert float(y1[7].abs().max()) == 0.0
    # strided output rows of a buffer view, in chunks
    buf = torch.zeros(3, 250, C, L, device="cuda")
    ops.odd_alias(z, 0, L, out=buf[:, 50:150], rowmax=rm["words"], rows_per_chunk=64)
    assert torch.equal(buf[:, 50:150].reshape(R, C, L), y0)
    # the partitioned convolution (more than 8193 taps: the reverb's impulse responses) leaves them too
    long = {}
    hl = torch.randn(6, 1, 20000, device="cuda") / 140.0
    zl = ops.fftconv(x[:6], ops.fir_spectrum(hl.reshape(6, 20000)), 20000, 1, Lout=L + 19999, rowmax=long)
    assert "words" in long
    assert torch.equal(long["words"].view(torch.float32).view(6, C), zl.abs().amax(-1))


@pytest.mark.parametrize("C,L,N,view", [(2, 8192, 4000, False), (1, 5000, 300, False), (2, 12001, 16384, True), (2, 4097, 64, True)])
def test_the_envelope_in_three_passes_equals_the_six_it_replaces(C, L, N, view):
    """A compressor whose smoother's convolve() aliases (upstream's default tap counts): energy -> truncated one-pole (full
    length) -> aliasing in double -> relu.  gfx_onepole_energy_f32 forms the energy inside the scan and leaves the rows'
    maxima, gfx_odd_alias_pair_precise_max_f32 takes them and clamps in its last pass: the same values as energy_kernel +
    onepole_kernel + the aliasing's own pass over the rows + torch.relu (the scan is the same arithmetic; the maxima are those
    of what was stored), poles from fast to the clamp (live truncation term), silence, strided buffer views."""
    from grafx_amd import ops
    from grafx_amd.processors.core.envelope import TruncatedOnePoleIIRFilter

    torch.manual_seed(L + N)
    R = 6
    if view:
        buf = torch.randn(2, 7, C, L, device="cuda")
        x = buf[:, 2:5]
    else:
        x = torch.randn(R, C, L, device="cuda")
    x[1] = 0
    z = torch.tensor([[0.0], [1.0], [-3.0], [6.0], [12.0], [3.0]], device="cuda")
    rm = {}
    full = ops.onepole_energy(x, z, N, Lout=L + N - 1, relu=False, rowmax=rm)
    want_full = ops.onepole(ops.energy(x), z, N, Lout=L + N - 1, relu=False)
    assert (full - want_full).abs().max() <= 2e-6 * want_full.abs().max()
    assert torch.equal(rm["words"].view(torch.float32), full.abs().amax(-1))
    assert (ops.onepole_energy(x, z, N) - ops.onepole(ops.energy(x), z, N)).abs().max() <= 2e-6 * want_full.abs().max()
    if (L + N - 1) % 2 == 1:
        a = ops.odd_alias(full, 0, L, precise=True, rowmax=rm["words"], relu=True)
        b = torch.relu(ops.odd_alias(full, 0, L, precise=True))
        assert torch.equal(a, b)
    m = TruncatedOnePoleIIRFilter(iir_len=N, flashfftconv=False).cuda()
    with torch.no_grad():
        assert (m.forward_energy(x, z) - m(ops.energy(x), z)).abs().max() <= 2e-6 * want_full.abs().max()
