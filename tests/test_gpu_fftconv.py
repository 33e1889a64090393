"""GPU parity: overlap-save FFT convolution kernels vs the CPU oracle (through the C ABI)."""
import pytest
import torch

from conftest import assert_close
from oracle import lti

pytestmark = pytest.mark.gpu


def _run(x, h, off, Lout):
    from grafx_amd import ops

    R, Cf, N = h.shape
    Hs = ops.fir_spectrum(h.cuda().reshape(R * Cf, N))
    return ops.fftconv(x.cuda(), Hs, N, Cf, Lout=Lout, off=off).cpu()


def _full(x, h):
    # float64 oracle of the full linear convolution
    return lti.linear_convolve(x.double(), h.double(), mode="full")


@pytest.mark.parametrize("L,N", [(1024, 128), (1000, 301), (40000, 4001), (16384, 1), (20000, 8193), (5, 3)])
@pytest.mark.parametrize("C,Cf", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_single_partition_causal(L, N, C, Cf):
    torch.manual_seed(L + N)
    x, h = torch.randn(3, C, L), torch.randn(3, Cf, N) / N**0.5
    y = _run(x, h, 0, L)
    assert_close(y, _full(x, h)[..., :L].float(), 1e-5, "causal")


@pytest.mark.parametrize("L,N", [(30000, 8200), (50000, 20001), (9000, 60001), (70000, 16385)])
def test_multi_partition_causal(L, N):
    torch.manual_seed(L + N)
    x, h = torch.randn(2, 2, L), torch.randn(2, 2, N) / N**0.5
    y = _run(x, h, 0, L)
    assert_close(y, _full(x, h)[..., :L].float(), 1e-5, "partitioned causal")


@pytest.mark.parametrize("C,Cf", [(1, 2), (2, 1), (1, 1)])
@pytest.mark.parametrize("L,N,mode", [(40000, 20001, "causal"), (33000, 8200, "zerophase"), (26000, 17000, "full"),
                                      (8192 * 3, 8192 * 2 + 1, "causal")])
def test_multi_partition_channel_broadcast_and_modes(C, Cf, L, N, mode):
    """The partitioned convolution (two output tiles per workgroup) with a mono signal through stereo filters and vice
    versa, in all three output modes, odd and even tile counts, against the float64 oracle."""
    torch.manual_seed(L + N + C + 2 * Cf)
    x, h = torch.randn(3, C, L), torch.randn(3, Cf, N) / N**0.5
    full = _full(x, h).float()
    off, Lout = {"causal": (0, L), "zerophase": (N // 2, L), "full": (0, L + N - 1)}[mode]
    assert_close(_run(x, h, off, Lout), full[..., off : off + Lout], 1e-5, f"partitioned {mode} C={C} Cf={Cf}")


@pytest.mark.parametrize("L,N", [(5000, 2047), (5001, 300), (20000, 9001)])
def test_zerophase_and_full(L, N):
    torch.manual_seed(7)
    x, h = torch.randn(2, 2, L), torch.randn(2, 1, N) / N**0.5
    full = _full(x, h).float()
    assert_close(_run(x, h, N // 2, L), full[..., N // 2 : N // 2 + L], 1e-5, "zerophase")
    assert_close(_run(x, h, 0, L + N - 1), full, 1e-5, "full")


def test_matches_reference_golden_even_P(golden):
    g = golden("g1_convolve")
    for (L, N) in [(1025, 128), (1024, 127)]:  # P even: reference == linear convolution
        x, h = g[f"x_L{L}_N{N}"], g[f"h_L{L}_N{N}"]
        assert_close(_run(x, h, 0, L), g[f"y_causal_L{L}_N{N}_C2_Cf2"], 1e-5, "golden causal")
        assert_close(_run(x, h, N // 2, L), g[f"y_zerophase_L{L}_N{N}_C2_Cf2"], 1e-5, "golden zerophase")


def test_strided_buffer_io():
    """Read from / write into slices of a (B, V, C, L) signal buffer in place."""
    from grafx_amd import ops

    torch.manual_seed(3)
    B, V, C, L, N = 3, 7, 2, 20000, 4001
    buf = torch.randn(B, V, C, L).cuda()
    h = (torch.randn(B * 2, 1, N) / N**0.5)
    Hs = ops.fir_spectrum(h.cuda().reshape(B * 2, N))
    src, dst = buf[:, 1:3], buf[:, 4:6]
    want = lti.linear_convolve(src.cpu().reshape(B * 2, C, L).double(), h.double(), "causal").float()
    before = buf.clone()
    ops.fftconv(src, Hs, N, 1, out=dst)
    assert_close(dst.cpu().reshape(B * 2, C, L), want, 1e-5, "strided")
    assert torch.equal(buf[:, :4], before[:, :4]) and torch.equal(buf[:, 6:], before[:, 6:])


def test_linearity_and_shift_at_full_size():
    """Size-independent properties at BASELINE scale (L = 2^17, N = 4001)."""
    torch.manual_seed(11)
    L, N = 131072, 4001
    x1, x2 = torch.randn(4, 2, L), torch.randn(4, 2, L)
    h = torch.randn(4, 1, N) / N**0.5
    y1, y2, y12 = _run(x1, h, 0, L), _run(x2, h, 0, L), _run(x1 + 2 * x2, h, 0, L)
    assert_close(y12, y1 + 2 * y2, 2e-6, "linearity")
    d = torch.zeros(4, 2, L)
    d[..., 5] = 1.0
    assert_close(_run(d, h, 0, L)[..., 5 : 5 + N], h.expand(4, 2, N), 2e-6, "impulse response")


@pytest.mark.gpu
@pytest.mark.parametrize("L,N,Cin,Cf", [(40000, 4001, 2, 1), (16385, 8193, 2, 2), (5000, 33, 1, 1)])
def test_fftconv_tee_copies_the_input_and_leaves_the_output_unchanged(L, N, Cin, Cf):
    """gfx_fftconv_tee_f32: same y as gfx_fftconv_f32, plus a bit-exact copy of x (strided destination rows)."""
    import torch

    from grafx_amd import ops

    torch.manual_seed(3)
    x = torch.randn(6, Cin, L, device="cuda")
    h = torch.randn(6, Cf, N, device="cuda") / N**0.5
    Hs = ops.fir_spectrum(h.view(-1, N))
    assert ops.fftconv_can_tee(Cin, Cf, L, L, 0, N)
    y0 = ops.fftconv(x, Hs, N, Cf)
    big = torch.full((2, 5, Cin, L), float("nan"), device="cuda")  # tee into rows 1..3 of a (B, n, C, L) buffer
    tee = big.narrow(1, 1, 3)
    y1 = ops.fftconv(x.view(2, 3, Cin, L), Hs, N, Cf, tee=tee)
    assert torch.equal(y0, y1)
    assert torch.equal(tee.reshape(6, Cin, L), x)
    assert torch.isnan(big[:, 0]).all() and torch.isnan(big[:, 4]).all()
    assert not ops.fftconv_can_tee(Cin, Cf, L, L, 1, N) and not ops.fftconv_can_tee(1, 2, L, L, 0, N)


@pytest.mark.gpu
@pytest.mark.parametrize("Nf,Lout", [(131072, 4001), (40000, 513), (20001, 8000)])
def test_long_partitions_give_the_same_convolution(Nf, Lout):
    """"Long filter, short output" (the filter gradient of a training step): gfx_fftconv_part_len picks partitions
    longer than 8192 taps (fewer of them, one output tile); the result must equal the default partitioning."""
    import torch

    from grafx_amd import ops

    torch.manual_seed(1)
    R, L = 3, 30000
    x = torch.randn(R, 2, L, device="cuda")
    h = torch.randn(R, 2, Nf, device="cuda") / Nf**0.5
    P = ops.part_len_for(Nf, Lout)
    assert P > 8192 and P % 2 == 0 and P + Lout <= 16384
    off = Nf - 1 - 17
    y0 = ops.fftconv(x, ops.fir_spectrum(h.view(-1, Nf)), Nf, 2, Lout=Lout, off=off)
    y1 = ops.fftconv(x, ops.fir_spectrum(h.view(-1, Nf), part_len=P), Nf, 2, Lout=Lout, off=off, part_len=P)
    assert (y0 - y1).abs().max() <= 2e-6 * y0.abs().max()
    assert ops.part_len_for(4001, 1000) == 0 and ops.part_len_for(60001, 131072) == 0   # default geometry cases
    if -(-Nf // P) != -(-Nf // 8192):  # different partition counts: spectra built for one geometry are refused by the other
        with pytest.raises(Exception):
            ops.fftconv(x, ops.fir_spectrum(h.view(-1, Nf), part_len=P), Nf, 2, Lout=Lout, off=off)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_CONV_SEEDS", 48))))
def test_random_geometries_match_the_oracle_convolution(seed):
    """Random signal / filter lengths (around the 8193-tap and 16384-sample tile boundaries too), output windows,
    channel broadcasts, shared filters and strided buffer views against the oracle's linear convolution."""
    import random

    import torch

    from grafx_amd import ops

    rng = random.Random(seed)
    torch.manual_seed(seed)
    L = rng.choice([1, 2, 3, 17, 1000, 4097, 12383, 12384, 12385, 16384, 16385, 40001, 70000])
    N = rng.choice([1, 2, 3, 64, 4001, 8191, 8192, 8193, 8194, 16383, 20000, 33000])
    Cin, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    B, n = rng.choice([(1, 1), (2, 3), (3, 2)])
    shared = rng.random() < 0.5
    full = L + N - 1
    off = rng.choice([0, N // 2, N - 1, rng.randint(0, N - 1)])
    Lout = rng.choice([L, full - off, max(1, min(L, 777))])
    Lout = max(1, min(Lout, full - off))
    buf = torch.randn(B, n + 2, Cin, L, device="cuda")
    x4 = buf.narrow(1, 1, n)                                  # strided (B, n, C, L) view
    h = torch.randn(n if shared else B * n, Cf, N, device="cuda") / max(N, 1) ** 0.5
    y = ops.fftconv(x4, ops.fir_spectrum(h.reshape(-1, N)), N, Cf, Lout=Lout, off=off, h_rows=h.shape[0])
    hx = (h.repeat(B, 1, 1) if shared else h).cpu()
    ref = lti.linear_convolve(x4.reshape(B * n, Cin, L).cpu(), hx, "full")[..., off : off + Lout]
    assert y.shape == ref.shape, (y.shape, ref.shape)
    scale = ref.abs().max().clamp_min(1e-6)
    assert (y.cpu() - ref).abs().max() <= 2e-5 * scale, f"L={L} N={N} C={Cin}/{Cf} off={off} Lout={Lout} shared={shared}"


@pytest.mark.parametrize("L,N", [(1, 1), (2, 5), (7, 3), (4001, 33), (8193, 100), (8194, 9), (20001, 4001),
                                 (131072, 4001), (40000, 1), (33333, 8193)])
def test_reversed_spectra_equal_the_spectra_of_the_flipped_copy(L, N):
    """gfx_fir_spectrum_rev_f32 (the filter-gradient correlation's "filter") reads the rows backwards in place:
    bit-identical to flipping first, for contiguous rows and for strided (B, n, C, L) views of a buffer."""
    from grafx_amd import ops

    def same(a, b):  # (filters x partitions, 17 slots, 256 threads, 4 floats); slot 16 is written by thread 0 only
        a, b = a.view(torch.float32).view(-1, 17, 256, 4), b.view(torch.float32).view(-1, 17, 256, 4)
        return torch.equal(a[:, :16], b[:, :16]) and torch.equal(a[:, 16, 0], b[:, 16, 0])

    torch.manual_seed(L + N)
    P = ops.part_len_for(L, N)
    x = torch.randn(3, 2, L, device="cuda")
    assert same(ops.fir_spectrum_reversed(x, part_len=P), ops.fir_spectrum(x.flip(-1).reshape(6, L), part_len=P))
    buf = torch.randn(3, 5, 2, L, device="cuda")
    view = buf.narrow(1, 1, 2)
    assert same(ops.fir_spectrum_reversed(view, part_len=P), ops.fir_spectrum(view.flip(-1).reshape(12, L), part_len=P))


@pytest.mark.parametrize("L,Lg,N,off,Cx,Cg", [(1, 1, 1, 0, 1, 1), (5, 5, 3, 0, 2, 2), (100, 100, 8, 4, 1, 2), (40000, 40000, 4001, 0, 2, 2),
                                               (40001, 40001, 4000, 2000, 2, 1), (12384, 12384, 4001, 0, 1, 1),
                                               (12385, 12385, 4001, 0, 2, 2), (30000, 38192, 8193, 0, 2, 2),
                                               (131072, 131072, 4001, 0, 2, 2), (9000, 9000, 2, 1, 2, 2)])
def test_filter_gradient_correlation_matches_float64(L, Lg, N, off, Cx, Cg):
    """gfx_fir_grad_f32: gh[k] = sum_n g[n] x[n + off - k] against a float64 FFT correlation (ragged tiles, odd
    lengths, channel broadcast, a 'full' gradient longer than the signal), contiguous rows and buffer views."""
    from grafx_amd import ops

    torch.manual_seed(L + N)
    R = 3
    x = torch.randn(R, Cx, L, device="cuda")
    g = torch.randn(R, Cg, Lg, device="cuda")

    def reference(x, g):
        P = L + Lg + N
        X = torch.fft.rfft(x.double(), n=P)
        G = torch.fft.rfft(g.double(), n=P)
        c = torch.fft.irfft(G * X.conj(), n=P)       # c[j] = sum_n g[n + j] x[n]  (circular, P long enough)
        k = torch.arange(N, device=x.device)
        return c[..., (k - off) % P]                 # gh[k] = sum_m x[m] g[m - off + k]

    want = reference(x, g).float()
    got = ops.fir_grad(x, g, N, off)
    assert got.shape == want.shape
    assert (got - want).abs().max() <= 2e-5 * want.abs().max()
    buf = torch.randn(R, 4, Cx, L, device="cuda")
    view = buf.narrow(1, 1, 2)
    g4 = torch.randn(R, 2, Cg, Lg, device="cuda")
    want4 = reference(view.reshape(-1, Cx, L), g4.reshape(-1, Cg, Lg)).float()
    got4 = ops.fir_grad(view, g4, N, off)
    assert (got4 - want4).abs().max() <= 2e-5 * want4.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_PIPE_SEEDS", 24))))
def test_pipe_schedule_matches_the_oracle_convolution(seed):
    """The hand-scheduled persistent kernel (gfx_fftconv_sched_f32, GFX_SCHED_PIPE: generated gfx950 assembly, csrc/asm)
    forced onto small random problems it covers -- tap counts of every overlap it is built for, even lengths around the tile
    boundaries (V = 12288), one to many tiles per row, fewer tiles than workgroups and more, channel broadcasts, shared
    filters, strided buffer views and the input copy -- against the oracle and the compiler-built tile kernel."""
    import random

    import torch

    from grafx_amd import ops

    rng = random.Random(4000 + seed)
    torch.manual_seed(seed)
    L = rng.choice([2, 18, 1000, 8192, 12286, 12288, 12290, 16384, 24576, 24578, 40002, 70000, 131072])
    N = rng.choice([2, 100, 513, 514, 1025, 2049, 3586, 4000, 4001, 4096, 4097, 7682, 8192, 8193])   # overlaps 512 .. 8192
    Cin, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    B, n = rng.choice([(1, 1), (2, 3), (3, 2), (5, 1), (7, 9)])
    shared = rng.random() < 0.5
    buf = torch.randn(B, n + 2, Cin, L, device="cuda")
    x4 = buf.narrow(1, 1, n)                                  # strided (B, n, C, L) view
    h = torch.randn(n if shared else B * n, Cf, N, device="cuda") / N ** 0.5
    Hs = ops.fir_spectrum(h.reshape(-1, N))
    tee = None
    if ops.fftconv_can_tee(Cin, Cf, L, L, 0, N) and rng.random() < 0.6:
        tee = torch.full((B, n, Cin, L), float("nan"), device="cuda")
    out = torch.full((B, n + 1, max(Cin, Cf), L), float("nan"), device="cuda")
    ops.fftconv(x4, Hs, N, Cf, out=out.narrow(1, 0, n), h_rows=h.shape[0], tee=tee, schedule="pipe")
    y = out[:, :n].reshape(B * n, max(Cin, Cf), L)
    hx = (h.repeat(B, 1, 1) if shared else h).cpu()
    ref = lti.linear_convolve(x4.reshape(B * n, Cin, L).cpu(), hx, "full")[..., :L]
    scale = ref.abs().max().clamp_min(1e-6)
    what = f"L={L} N={N} C={Cin}/{Cf} B={B} n={n} shared={shared} tee={tee is not None}"
    assert torch.isfinite(y).all(), what
    assert torch.isnan(out[:, n]).all(), what                # the row behind the stage's rows is not touched
    assert (y.cpu() - ref).abs().max() <= 2e-5 * scale, what
    if tee is not None:
        assert torch.equal(tee, x4), what
    y_tile = ops.fftconv(x4, Hs, N, Cf, h_rows=h.shape[0], schedule="tile")
    assert (y - y_tile).abs().max() <= 4e-6 * scale, what


@pytest.mark.gpu
def test_pipe_schedule_headline_shape_against_the_tile_schedule():
    """The shape the console's first stage launches (32 shared filters, stereo, L = 131072, N = 4001, output and input
    copy written into strided views of one buffer) at 8 graphs: persistent kernel vs one tile per workgroup, every
    sample; and what the schedule does not cover is refused by name."""
    import torch

    from grafx_amd import ops

    torch.manual_seed(5)
    B, n, L, N = 8, 32, 131072, 4001
    x4 = torch.randn(B, n, 2, L, device="cuda")
    h = torch.randn(n, 1, N, device="cuda") / N ** 0.5
    Hs = ops.fir_spectrum(h.reshape(-1, N))
    buf = torch.zeros(B, 3 * n, 2, L, device="cuda")
    ops.fftconv(x4, Hs, N, 1, out=buf.narrow(1, n, n), tee=buf.narrow(1, 0, n), h_rows=n, schedule="pipe")
    want = ops.fftconv(x4, Hs, N, 1, h_rows=n, schedule="tile")
    assert torch.equal(buf[:, :n], x4)
    assert (buf[:, n : 2 * n].reshape(B * n, 2, L) - want).abs().max() <= 4e-6 * want.abs().max()
    assert (buf[:, 2 * n :] == 0).all()
    with pytest.raises(RuntimeError):                         # odd length: not covered, refused
        ops.fftconv(x4[..., :4097], Hs, N, 1, h_rows=n, schedule="pipe")
    h2 = torch.randn(n, 1, 1100, device="cuda")
    with pytest.raises(RuntimeError):                         # no variant for a 1536-sample overlap
        ops.fftconv(x4, ops.fir_spectrum(h2.reshape(-1, 1100)), 1100, 1, h_rows=n, schedule="pipe")


@pytest.mark.parametrize("L,N,off", [(30000, 8200, 0), (50000, 20001, 0), (9000, 60001, 0), (70000, 16385, 0),
                                     (240000, 60001, 0), (41000, 20001, 10000), (65536, 30000, 29999), (8192, 9000, 0),
                                     (131072, 60000, 0), (24577, 8194, 3)])
def test_two_output_tiles_per_workgroup_equal_the_one_tile_schedule(L, N, off):
    """macinv_pair_kernel (GFX_SCHED_AUTO for partitioned filters: two consecutive output tiles per 512-thread workgroup,
    each group of 256 threads multiplying half the mirrored bin pairs of both tiles) against macinv_kernel (GFX_SCHED_TILE):
    the same partitions added in the same order -- equal to a few units in the last place of the largest output (the
    twiddle arithmetic of the two kernels is contracted differently), including odd tile counts (a last group of one
    tile), windows before the row start / past its end, output offsets, shared filter rows and strided views."""
    from grafx_amd import ops

    torch.manual_seed(L + N)
    B, n = 2, 3
    buf = torch.randn(B, n + 1, 2, L, device="cuda")
    x4 = buf.narrow(1, 1, n)
    h = torch.randn(n, 2, N, device="cuda") / N**0.5
    Hs = ops.fir_spectrum(h.reshape(-1, N))
    Lout = min(L, L + N - 1 - off)
    a = ops.fftconv(x4, Hs, N, 2, Lout=Lout, off=off, h_rows=n, schedule="auto")
    b = ops.fftconv(x4, Hs, N, 2, Lout=Lout, off=off, h_rows=n, schedule="tile")
    assert (a - b).abs().max().item() <= 4e-7 * b.abs().max().item()
    assert torch.equal(a, ops.fftconv(x4, Hs, N, 2, Lout=Lout, off=off, h_rows=n, schedule="auto"))
    ref = lti.linear_convolve(x4.reshape(B * n, 2, L).cpu(), h.repeat(B, 1, 1).cpu(), "full")[..., off : off + Lout]
    assert_close(a.cpu(), ref, 1e-5, "partitioned, two tiles per workgroup")
    assert_close(b.cpu(), ref, 1e-5, "partitioned, one tile per workgroup")
