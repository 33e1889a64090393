"""Parity tests of the round-2 experimental schedules (ping-pong, half-exchange, wide), kept with the kernels they
tested; they ran green in round 2 (GPUTEST_r02) and are not collected any more."""
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_PP_SEEDS", 40))))
def test_pingpong_schedule_matches_the_oracle_convolution(seed):
    """The persistent ping-pong kernel (gfx_fftconv_sched_f32, GFX_SCHED_PINGPONG) forced onto small random problems:
    signal / filter lengths around its tile boundaries (V = 16384 - roundup(N - 1, 512)), one to many tiles per row,
    odd stream lengths, channel broadcasts, shared filters, strided views, output windows and the input copy."""
    import random

    import torch

    from grafx_amd import ops

    rng = random.Random(1000 + seed)
    torch.manual_seed(seed)
    L = rng.choice([1, 2, 3, 17, 1000, 8191, 8192, 8193, 12287, 12288, 12289, 16384, 16385, 24577, 40001, 70000])
    N = rng.choice([1, 2, 3, 64, 511, 512, 513, 514, 4001, 8191, 8192, 8193])
    Cin, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    B, n = rng.choice([(1, 1), (2, 3), (3, 2), (5, 1)])
    shared = rng.random() < 0.5
    full = L + N - 1
    off = rng.choice([0, 0, N // 2, N - 1, rng.randint(0, N - 1)])
    Lout = rng.choice([L, full - off, max(1, min(L, 777))])
    Lout = max(1, min(Lout, full - off))
    buf = torch.randn(B, n + 2, Cin, L, device="cuda")
    x4 = buf.narrow(1, 1, n)                                  # strided (B, n, C, L) view
    h = torch.randn(n if shared else B * n, Cf, N, device="cuda") / max(N, 1) ** 0.5
    Hs = ops.fir_spectrum(h.reshape(-1, N))
    tee = None
    if ops.fftconv_can_tee(Cin, Cf, L, Lout, off, N) and rng.random() < 0.7:
        tee = torch.full((B, n, Cin, L), float("nan"), device="cuda")
    y = ops.fftconv(x4, Hs, N, Cf, Lout=Lout, off=off, h_rows=h.shape[0], tee=tee, schedule="pingpong")
    hx = (h.repeat(B, 1, 1) if shared else h).cpu()
    ref = lti.linear_convolve(x4.reshape(B * n, Cin, L).cpu(), hx, "full")[..., off : off + Lout]
    assert y.shape == ref.shape, (y.shape, ref.shape)
    scale = ref.abs().max().clamp_min(1e-6)
    what = f"L={L} N={N} C={Cin}/{Cf} B={B} n={n} off={off} Lout={Lout} shared={shared} tee={tee is not None}"
    assert torch.isfinite(y).all(), what
    assert (y.cpu() - ref).abs().max() <= 2e-5 * scale, what
    if tee is not None:
        assert torch.equal(tee, x4), what
    y_tile = ops.fftconv(x4, Hs, N, Cf, Lout=Lout, off=off, h_rows=h.shape[0], schedule="tile")
    assert (y - y_tile).abs().max() <= 4e-6 * scale, what
    # the half-exchange schedule (three workgroups per CU) computes the same tiles as the tile schedule
    tee2 = None if tee is None else torch.full_like(tee, float("nan"))
    y_hx = ops.fftconv(x4, Hs, N, Cf, Lout=Lout, off=off, h_rows=h.shape[0], tee=tee2, schedule="halfx")
    assert (y_hx - y_tile).abs().max() <= 2e-6 * scale, what
    if tee2 is not None:
        assert torch.equal(tee2, x4), what


@pytest.mark.gpu
def test_pingpong_schedule_headline_shape_against_the_tile_schedule():
    """The shape the console's first stage launches (32 shared filters, stereo, L = 131072, N = 4001, output and input
    copy written into strided views of one buffer) at 8 graphs: ping-pong vs one-tile-per-workgroup, every sample."""
    import torch

    from grafx_amd import ops

    torch.manual_seed(5)
    B, n, L, N = 8, 32, 131072, 4001
    x4 = torch.randn(B, n, 2, L, device="cuda")
    h = torch.randn(n, 1, N, device="cuda") / N ** 0.5
    Hs = ops.fir_spectrum(h.reshape(-1, N))
    buf = torch.zeros(B, 3 * n, 2, L, device="cuda")
    ops.fftconv(x4, Hs, N, 1, out=buf.narrow(1, n, n), tee=buf.narrow(1, 0, n), h_rows=n, schedule="pingpong")
    want = ops.fftconv(x4, Hs, N, 1, h_rows=n, schedule="tile")
    assert torch.equal(buf[:, :n], x4)
    assert (buf[:, n : 2 * n].reshape(B * n, 2, L) - want).abs().max() <= 4e-6 * want.abs().max()
    assert (buf[:, 2 * n :] == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("GRAFX_FUZZ_WIDE_SEEDS", 30))))
def test_wide_schedule_matches_the_oracle_convolution(seed):
    """The 512-thread tile (gfx_fftconv_sched_f32, GFX_SCHED_WIDE: 16 points per thread, radix-32 pass split over lane
    pairs, one-output spectral product) on random causal problems: lengths around its tile boundaries, channel
    broadcasts, shared filters, strided views and the input copy -- against the oracle and the 256-thread tile."""
    import random

    import torch

    from grafx_amd import ops

    rng = random.Random(7000 + seed)
    torch.manual_seed(seed)
    L = rng.choice([2, 18, 1000, 8192, 12286, 12288, 12290, 16384, 16386, 24578, 40002, 70000])
    N = rng.choice([1, 2, 3, 64, 511, 512, 513, 514, 4001, 8191, 8192, 8193])
    Cin, Cf = rng.choice([(1, 1), (2, 1), (1, 2), (2, 2)])
    B, n = rng.choice([(1, 1), (2, 3), (3, 2), (5, 1)])
    shared = rng.random() < 0.5
    Lout = rng.choice([L, L, max(2, min(L, 776))])
    buf = torch.randn(B, n + 2, Cin, L, device="cuda")
    x4 = buf.narrow(1, 1, n)
    h = torch.randn(n if shared else B * n, Cf, N, device="cuda") / max(N, 1) ** 0.5
    Hs = ops.fir_spectrum(h.reshape(-1, N))
    tee = None
    if ops.fftconv_can_tee(Cin, Cf, L, Lout, 0, N) and rng.random() < 0.7:
        tee = torch.full((B, n, Cin, L), float("nan"), device="cuda")
    y = ops.fftconv(x4, Hs, N, Cf, Lout=Lout, h_rows=h.shape[0], tee=tee, schedule="wide")
    hx = (h.repeat(B, 1, 1) if shared else h).cpu()
    ref = lti.linear_convolve(x4.reshape(B * n, Cin, L).cpu(), hx, "full")[..., :Lout]
    scale = ref.abs().max().clamp_min(1e-6)
    what = f"L={L} N={N} C={Cin}/{Cf} B={B} n={n} Lout={Lout} shared={shared} tee={tee is not None}"
    assert torch.isfinite(y).all(), what
    assert (y.cpu() - ref).abs().max() <= 2e-5 * scale, what
    if tee is not None:
        assert torch.equal(tee, x4), what
    y_tile = ops.fftconv(x4, Hs, N, Cf, Lout=Lout, h_rows=h.shape[0], schedule="tile")
    assert (y - y_tile).abs().max() <= 4e-6 * scale, what


@pytest.mark.gpu
def test_wide_schedule_rejects_what_it_does_not_cover():
    import torch

    from grafx_amd import ops

    x = torch.randn(2, 1, 4097, device="cuda")
    h = torch.randn(2, 1, 100, device="cuda")
    Hs = ops.fir_spectrum(h.reshape(-1, 100))
    with pytest.raises(RuntimeError):
        ops.fftconv(x, Hs, 100, 1, schedule="wide")          # odd length
    with pytest.raises(RuntimeError):
        ops.fftconv(x[..., :4096], Hs, 100, 1, off=50, schedule="wide")   # output offset
