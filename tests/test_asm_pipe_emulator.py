"""CPU check of the hand-scheduled convolution kernel (grafx_amd/csrc/asm): the generated instruction list is executed by
the wave-level emulator of `isa.py` -- the same list that is printed as gfx950 assembly -- on small problems, against a
float64 convolution.  The emulator also rejects a schedule that reads a register before its load was waited for, or
hands data across waves through LDS without a barrier, so a pass here means the arithmetic, the register map, the
software pipeline across tiles (prologue, alternating banks, both exits) and the automatic s_waitcnt placement agree.

No GPU involved; the GPU parity tests of the assembled kernel are in tests/test_gpu_fftconv.py."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def _twiddle_table():
    t = np.arange(256)
    rows = []
    for row in range(20):
        if row < 4:
            num, den = t * row, 8192.0
        elif row < 12:
            num, den = t * 4 * (row - 4), 8192.0
        elif row < 16:
            num, den = (t & 15) * (row - 12), 256.0
        else:
            num, den = (t & 15) * 4 * (row - 16), 256.0
        ang = 2 * np.pi * num / den
        rows.append(np.stack([np.cos(ang), -np.sin(ang)], -1))
    return np.stack(rows).astype(np.float32)          # [20][256][2]


def _spectra(h):
    """taps (F, N) -> the (He, Ho) thread layout hspec_kernel writes: [F][17][256][4] float32"""
    import fft_tile_model as model

    out = np.zeros((h.shape[0], 17, 256, 4), np.float32)
    for f in range(h.shape[0]):
        He, Ho = model.filter_slots(h[f].astype(np.float64))
        out[f, :, :, 0], out[f, :, :, 1] = He.real, He.imag
        out[f, :, :, 2], out[f, :, :, 3] = Ho.real, Ho.imag
    return out


def _run(tee, B, n, C, Cf, L, N, hrows, tiles_per_wg=None, seed=0, rowmax_out=None, Lout=None, **knobs):
    from grafx_amd.csrc.asm import gen_fftconv_pipe as gen
    from grafx_amd.csrc.asm.isa import Buffer, Emulator

    rng = np.random.default_rng(seed)
    O = (N - 1 + 511) & ~511
    V = 16384 - O
    a_lo = O // 512
    Cout = max(C, Cf)
    R = B * n
    Lout = L if Lout is None else Lout
    ntiles = (Lout + V - 1) // V
    nblocks = R * Cout * ntiles
    prog = gen.PipeGen(tee, a_lo, **knobs).build()
    # signals live in a (B, nodes, C, L) buffer with more node rows than the stage uses (strided views, as in the render)
    nodes = n + 3
    xbuf = rng.standard_normal((B, nodes, C, L)).astype(np.float32)
    ybuf = np.full((B, nodes, Cout, Lout), np.nan, np.float32)
    cbuf = np.full((B, nodes, C, L), np.nan, np.float32)
    h = (rng.standard_normal((hrows * Cf, N)) / np.sqrt(N)).astype(np.float32)
    mem = Buffer()
    xa, ya, ca = mem.alloc(xbuf), mem.alloc(ybuf), mem.alloc(cbuf)
    ha, ta = mem.alloc(_spectra(h)), mem.alloc(_twiddle_table())
    x0 = xa + 4 * (1 * C * L)            # the stage reads node rows 1 .. n
    y0 = ya + 4 * (2 * Cout * Lout)      # and writes node rows 2 .. n + 1
    rm = {}
    if rowmax_out is not None:           # one word per output row-channel, zeroed by the caller (the rm_* descriptor)
        rma = mem.alloc(np.zeros(R * Cout + 3, np.uint32).view(np.float32))
        rm = dict(rm_lo=rma & 0xFFFFFFFF, rm_hi=(rma >> 32) & 0xFFFF, rm_rec=4 * R * Cout, rm_flags=gen.RSRC_FLAGS)
    c0 = ca + 4 * (0 * C * L)
    per_xcd = nblocks if tiles_per_wg is None else tiles_per_wg
    m = {k: gen.magic(d) for k, d in (("ntiles", ntiles), ("inner", n), ("hrows", hrows))}
    args = gen.pack_args(
        x_lo=x0, x_hi=x0 >> 32, h_lo=ha, h_hi=ha >> 32, y_lo=y0, y_hi=y0 >> 32, c_lo=c0, c_hi=c0 >> 32, tw_lo=ta, tw_hi=ta >> 32,
        L_bytes=4 * L, Lout_bytes=4 * Lout, V_bytes=4 * V, O_bytes=4 * O, ntiles=ntiles, nblocks=nblocks,
        m_ntiles=m["ntiles"][0], sh_ntiles=m["ntiles"][1], inner=n, m_inner=m["inner"][0], sh_inner=m["inner"][1],
        hrows=hrows, m_hrows=m["hrows"][0], sh_hrows=m["hrows"][1], cout_shift=Cout - 1, cout_mask=Cout - 1,
        cin_mask=0 if C == 1 else 1, cf_mask=0 if Cf == 1 else 1, Cf=Cf, per_xcd=per_xcd, wgs_per_xcd=1,
        xs_outer_lo=4 * nodes * C * L, xs_inner=4 * C * L, xs_ch=4 * L,
        ys_outer_lo=4 * nodes * Cout * Lout, ys_inner=4 * Cout * Lout, ys_ch=4 * Lout,
        cs_outer_lo=4 * nodes * C * L, cs_inner=4 * C * L, cs_ch=4 * L, **rm)
    nwg = (nblocks + per_xcd - 1) // per_xcd      # consecutive runs of `per_xcd` tiles, one workgroup each
    for wg in range(nwg):
        Emulator(prog, mem, gen.TILE_LDS_BYTES, kernarg=args, wg_id=wg, rng=np.random.default_rng(100 + wg)).run()
    y = mem.read_back(ya).reshape(ybuf.shape)[:, 2: 2 + n]
    cc = mem.read_back(ca).reshape(cbuf.shape)[:, 0: n]
    x = xbuf[:, 1: 1 + n]
    want = np.zeros((B, n, Cout, Lout))
    for b in range(B):
        for j in range(n):
            r = b * n + j
            for c in range(Cout):
                hh = h[(r % hrows) * Cf + (c if Cf == 2 else 0)].astype(np.float64)
                want[b, j, c] = np.convolve(x[b, j, c if C == 2 else 0].astype(np.float64), hh)[:Lout]
    if rowmax_out is not None:
        rowmax_out.append(mem.read_back(rma).view(np.uint32))
    untouched = np.concatenate([mem.read_back(ya).reshape(ybuf.shape)[:, :2].ravel(),
                                mem.read_back(ya).reshape(ybuf.shape)[:, 2 + n:].ravel()])
    return y, want, cc, x, untouched


@pytest.mark.parametrize("tee,C,Cf,L,tiles_per_wg", [(True, 1, 1, 20002, None), (False, 2, 1, 12288 * 2, 2), (True, 2, 2, 9000, 3)])
def test_pipe_kernel_in_the_emulator(tee, C, Cf, L, tiles_per_wg):
    B, n, N = 2, 2, 4001
    y, want, cc, x, untouched = _run(tee, B, n, C, Cf, L, N, hrows=n, tiles_per_wg=tiles_per_wg)
    assert np.isfinite(y).all()
    err = np.abs(y - want).max() / np.abs(want).max()
    assert err < 5e-6, err
    assert np.isnan(untouched).all()                 # rows of the buffer the stage does not own stay untouched
    if tee:
        assert np.array_equal(cc, x)
    else:
        assert np.isnan(cc).all()


@pytest.mark.parametrize("C,Cf,L,N,tiles_per_wg", [(2, 1, 30000, 4001, 2), (1, 2, 12288 * 2 - 5, 4001, None), (2, 2, 9000, 513, 1)])
def test_pipe_kernel_leaves_the_rows_maxima(C, Cf, L, N, tiles_per_wg):
    """(The generator's `rowmax` knob: not in the shipped build -- this kernel needs an even output length, the convolution in
    front of the odd-length aliasing has an odd one and runs on fftconv1_kernel, which takes the maxima itself -- kept and
    tested as a generator capability.)  A full-length convolution with the rm_* descriptor set: one
    word per output row-channel receives the bits of max |y| over the row -- whatever the split of the row's tiles over
    workgroups (atomic maximum) -- to within the rounding noise of the samples past the row's end that a tile also
    computes (exact zeros in exact arithmetic); rows of zeros give zero; the words behind the buffer stay untouched."""
    B, n = 2, 2
    got = []
    y, want, _, _, _ = _run(False, B, n, C, Cf, L, N, hrows=n, tiles_per_wg=tiles_per_wg, rowmax_out=got, Lout=L + N - 1, rowmax=1)
    assert np.abs(y - want).max() / np.abs(want).max() < 5e-6
    words = got[0]
    Cout = max(C, Cf)
    mx = np.abs(y.reshape(B * n * Cout, -1)).max(1)
    kernel = words[: B * n * Cout].view(np.float32)
    assert (kernel >= mx).all() and (kernel <= mx * (1 + 1e-6) + 1e-6 * np.abs(y).max()).all(), (kernel, mx)
    assert (words[B * n * Cout:] == 0).all()


@pytest.mark.parametrize("knobs", [dict(place="barrier"), dict(place="after"), dict(fwd_window=(0.0, 1.0), inv_window=(0.0, 1.0))])
def test_pipe_kernel_schedule_knobs_keep_the_results(knobs):
    """Where the memory instructions sit is a tuning knob of the generator; the automatic s_waitcnt pass has to keep every
    placement correct."""
    y, want, cc, x, _ = _run(True, 1, 3, 2, 1, 12288 + 4096, 4001, hrows=3, tiles_per_wg=4, **knobs)
    assert np.abs(y - want).max() / np.abs(want).max() < 5e-6
    assert np.array_equal(cc, x)


@pytest.mark.parametrize("N,tee", [(513, True), (8193, False), (1100, False)])
def test_pipe_kernel_other_overlap_variants(N, tee):
    """the kernel is generated per overlap (a_lo = O / 512 rows of a tile are not stored): the shortest, the longest and
    a middle one"""
    y, want, cc, x, _ = _run(tee, 1, 2, 1, 1, 30000, N, hrows=2, tiles_per_wg=3)
    assert np.abs(y - want).max() / np.abs(want).max() < 5e-6
    if tee:
        assert np.array_equal(cc, x)


def _run_corr(B, n, Cx, Cg, L, N, seed=0, **knobs):
    from grafx_amd.csrc.asm import gen_corr_pipe as gen
    from grafx_amd.csrc.asm.gen_fftconv_pipe import TILE_LDS_BYTES, magic
    from grafx_amd.csrc.asm.isa import Buffer, Emulator

    rng = np.random.default_rng(seed)
    Cout = max(Cx, Cg)
    R = B * n
    V = 16384 - (N & ~1)
    ntiles = (L + V - 1) // V
    nodes = n + 2
    xbuf = rng.standard_normal((B, nodes, Cx, L)).astype(np.float32)
    gbuf = rng.standard_normal((B, nodes, Cg, L)).astype(np.float32)
    out = np.full((R * Cout, N), np.nan, np.float32)
    mem = Buffer()
    xa, ga, oa, ta = mem.alloc(xbuf), mem.alloc(gbuf), mem.alloc(out), mem.alloc(_twiddle_table())
    x0, g0 = xa + 4 * (1 * Cx * L), ga + 4 * (2 * Cg * L)        # the stage's rows: nodes 1..n of x, 2..n+1 of g
    prog = gen.CorrGen(**knobs).build()
    nblocks = R * Cout
    grid = (nblocks + 7) & ~7
    m_in = magic(n)
    tail = N - 1 if N & 1 else None
    args = gen.pack_args(
        x_lo=x0, x_hi=x0 >> 32, g_lo=g0, g_hi=g0 >> 32, o_lo=oa, o_hi=oa >> 32, tw_lo=ta, tw_hi=ta >> 32,
        L_bytes=4 * L, Lg_bytes=4 * L, V_bytes=4 * V, N_even_bytes=4 * (N & ~1), ntiles=ntiles, nblocks=nblocks,
        inner=n, m_inner=m_in[0], sh_inner=m_in[1], cout_shift=Cout - 1, cout_mask=Cout - 1,
        cx_mask=0 if Cx == 1 else 1, cg_mask=0 if Cg == 1 else 1, out_row_bytes=4 * N,
        tail_row=255 if tail is None else (tail // 2) >> 8, tail_lane=0 if tail is None else (tail // 2) & 255,
        tail_off=0 if tail is None else 2048 * ((tail // 2) >> 8),
        scale=int(np.float32(1.0 / (4 * 8192)).view(np.uint32)), pad0=grid // 8,
        xs_outer_lo=4 * nodes * Cx * L, xs_inner=4 * Cx * L, xs_ch=4 * L,
        gs_outer_lo=4 * nodes * Cg * L, gs_inner=4 * Cg * L, gs_ch=4 * L)
    for wg in range(grid):
        Emulator(prog, mem, TILE_LDS_BYTES, kernarg=args, wg_id=wg, rng=np.random.default_rng(7 + wg)).run()
    got = mem.read_back(oa).reshape(R, Cout, N)
    want = np.zeros((R, Cout, N))
    for b in range(B):
        for j in range(n):
            for c in range(Cout):
                xr = xbuf[b, 1 + j, c if Cx == 2 else 0].astype(np.float64)
                gr = gbuf[b, 2 + j, c if Cg == 2 else 0].astype(np.float64)
                full = np.correlate(gr, xr, "full")          # full[L - 1 + k] = sum_n g[n + k] x[n] ... lag k of g against x
                want[b * n + j, c] = full[L - 1: L - 1 + N]
    return got, want


@pytest.mark.parametrize("Cx,Cg,L,N", [(1, 1, 20002, 4001), (2, 1, 13000, 4000), (1, 2, 9000, 513)])
def test_corr_kernel_in_the_emulator(Cx, Cg, L, N):
    """gfx_corr_pipe (the filter-gradient correlation, csrc/asm/gen_corr_pipe.py): gh[k] = sum_n g[n] x[n - k]"""
    got, want = _run_corr(1, 2, Cx, Cg, L, N)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() / np.abs(want).max() < 5e-6


def _tiny_launch(prog):
    """one workgroup, two tiles, of the convolution kernel on `prog`"""
    from grafx_amd.csrc.asm import gen_fftconv_pipe as gen
    from grafx_amd.csrc.asm.isa import Buffer, Emulator

    rng = np.random.default_rng(5)
    L, N, O, V = 12288 * 2, 4001, 4096, 12288
    x = rng.standard_normal((1, 1, L)).astype(np.float32)
    y = np.zeros_like(x)
    h = (rng.standard_normal((1, N)) / 60).astype(np.float32)
    mem = Buffer()
    xa, ya, ha, ta = mem.alloc(x), mem.alloc(y), mem.alloc(_spectra(h)), mem.alloc(_twiddle_table())
    one = gen.magic(1)
    two = gen.magic(2)
    args = gen.pack_args(x_lo=xa, x_hi=xa >> 32, h_lo=ha, h_hi=ha >> 32, y_lo=ya, y_hi=ya >> 32, tw_lo=ta, tw_hi=ta >> 32,
                         L_bytes=4 * L, Lout_bytes=4 * L, V_bytes=4 * V, O_bytes=4 * O, ntiles=2, nblocks=2,
                         m_ntiles=two[0], sh_ntiles=two[1], inner=1, m_inner=one[0], sh_inner=one[1], hrows=1,
                         m_hrows=one[0], sh_hrows=one[1], Cf=1, per_xcd=2, wgs_per_xcd=1,
                         xs_inner=4 * L, xs_ch=4 * L, ys_inner=4 * L, ys_ch=4 * L)
    Emulator(prog, mem, gen.TILE_LDS_BYTES, kernarg=args, wg_id=0, rng=np.random.default_rng(1)).run()
    return mem.read_back(ya).reshape(-1), np.convolve(x.reshape(-1).astype(np.float64), h.reshape(-1).astype(np.float64))[:L]


def test_the_emulator_rejects_a_schedule_with_a_missing_wait_or_barrier():
    """The checker has to be able to fail: drop one s_waitcnt that guards a loaded register, or one s_barrier between an
    LDS write and another wave's read, from the generated program and the emulator must refuse it (and the intact
    program must pass)."""
    from grafx_amd.csrc.asm import gen_fftconv_pipe as gen
    from grafx_amd.csrc.asm.isa import EmuError, Label

    prog = gen.PipeGen(False, 8).build()
    got, want = _tiny_launch(prog)
    assert np.abs(got - want).max() / np.abs(want).max() < 5e-6
    loop = next(i for i, p in enumerate(prog) if isinstance(p, Label) and p.name == ".Lloop")
    # the first vmcnt wait inside the loop (it guards the window / spectrum registers)
    k = next(i for i in range(loop, len(prog)) if not isinstance(prog[i], Label) and prog[i].op == "s_waitcnt"
             and prog[i].mods.get("vmcnt") is not None)
    with pytest.raises(EmuError, match="outstanding"):
        _tiny_launch(prog[:k] + prog[k + 1:])
    # the first barrier of the loop (between the S1 writes of pass 1 and the reads of pass 2)
    b = next(i for i in range(loop, len(prog)) if not isinstance(prog[i], Label) and prog[i].op == "s_barrier")
    with pytest.raises(EmuError, match="barrier"):
        _tiny_launch(prog[:b] + prog[b + 1:])
