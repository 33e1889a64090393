"""The compressor's smoother with a LONG memory on the tile grid (csrc/dynamics.hip, "look-back" tiles, round 5).

Round 4's dependency-free tiles took a row only if its smoother forgot within 256 samples (a <= 0.898); every slower
pole went to one workgroup per row, and in a fused routing sum those rows were written and read back.  Now a tile whose
row remembers up to 64 tiles (truncation term dead: a^iir_len <= 1e-12) gets its entry state from the published
aggregates of the tiles before it.  Checked here: against the oracle (Compressor / NoiseGate, reference
dynamics.py:361-489, core/envelope.py:34-60) with the float64 tie-breaker; against round 4's row kernel on the same
rows (`ops.DYN_LOOKBACK = False`); with rows of all three kinds in one launch; through the fused routing sum; repeated
launches on a recycled workspace (the granules of the previous launch must not be seen)."""
import pytest
import torch

import oracle
from conftest import assert_parity

pytestmark = pytest.mark.gpu


def _params(R, z, seed):
    g = torch.Generator().manual_seed(seed)
    return {"log_threshold": torch.randn(R, 1, generator=g) - 1, "log_ratio": torch.randn(R, 1, generator=g),
            "log_knee": torch.randn(R, 1, generator=g), "z_alpha_pre": z}


@pytest.mark.parametrize("L,iir_len", [(131072, 16383), (20000, 16384), (131072, 4001)])
def test_long_pole_compressor_matches_the_oracle(L, iir_len):
    from grafx_amd.processors import Compressor

    # poles 0.953 .. 0.9982: H = 570 .. 15 000 samples (2 .. 30 tiles); at iir_len 4001 the slower ones keep a live truncation
    # term and stay on the row kernel -- all kinds in one call
    z = torch.tensor([[3.0], [4.0], [5.0], [5.7], [6.0], [6.3], [0.0], [-2.0]])
    R = z.shape[0]
    torch.manual_seed(L + iir_len)
    x = torch.randn(R, 2, L) * torch.linspace(0.5, 1.0, R)[:, None, None]
    x[:, :, L // 2 : L // 2 + 3000] *= 1.5                       # a burst: the envelope has something to remember
    p = _params(R, z, 3)
    m = Compressor(energy_smoother="iir", iir_len=iir_len, flashfftconv=False).cuda()
    with torch.no_grad():
        y = m(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
    o = oracle.OracleCompressor(iir_len=iir_len)
    ref = o(x, **p)
    ref64 = o(x.double(), **{k: v.double() for k, v in p.items()}).float()
    assert_parity(y, ref, ref64, 1e-5, f"long-pole compressor L={L} iir_len={iir_len}")


@pytest.mark.parametrize("knee,gate", [("quadratic", False), ("hard", True), ("exponential", False)])
def test_look_back_tiles_equal_the_row_kernel(knee, gate):
    """Same rows through the look-back tiles and through round 4's one-workgroup-per-row scan: two summation orders of the
    same linear recursion."""
    from grafx_amd import ops

    R, L = 96, 131072
    torch.manual_seed(7)
    x = torch.randn(R, 2, L, device="cuda") * 0.5
    z = torch.linspace(2.5, 6.3, R, device="cuda")[:, None]
    lt, lr, lk = (torch.randn(R, 1, device="cuda") for _ in range(3))
    old = ops.DYN_LOOKBACK
    try:
        ops.DYN_LOOKBACK = True
        a = ops.dynamics_fused(x, lt, lr, lk, z, smoother=1, iir_len=16383, knee=knee, gate=gate)
        ops.DYN_LOOKBACK = False
        b = ops.dynamics_fused(x, lt, lr, lk, z, smoother=1, iir_len=16383, knee=knee, gate=gate)
    finally:
        ops.DYN_LOOKBACK = old
    assert (a - b).abs().max() <= 2e-6 * b.abs().max(), float((a - b).abs().max() / b.abs().max())


def test_recycled_workspace_and_repeatability():
    """Launch after launch on memory the caching allocator hands back: the granules of an earlier launch (other data, same
    addresses) must never be taken for this launch's; and the same input gives the same bits every time."""
    from grafx_amd import ops

    R, L = 40, 65536
    z = torch.full((R, 1), 6.0, device="cuda")
    lt, lr, lk = (torch.zeros(R, 1, device="cuda") for _ in range(3))
    outs = []
    for it in range(6):
        torch.manual_seed(it % 2)
        x = torch.randn(R, 2, L, device="cuda") * (1.0 + it % 2)
        y = ops.dynamics_fused(x, lt, lr, lk, z, smoother=1, iir_len=16383, knee="quadratic", gate=False)
        outs.append(y.clone())
        del x, y
    for it in range(2, 6):
        assert torch.equal(outs[it], outs[it % 2]), it
    assert not torch.equal(outs[0], outs[1])


def test_console_render_with_long_poles_matches_the_oracle():
    """The fused routing sum (dyn_oneshot_mix_kernel) with look-back rows: a small console rendered through render_grafx
    against the oracle processors, compressor poles at 0.9975."""
    import bench
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    B, L = 2, 16384
    lens = dict(fsm_fir_len=513, iir_len=4095, ir_len=3001)
    G = bench.console_graph()
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    rd_dev = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")   # (.to moves in place)
    procs = {k: v.cuda() for k, v in bench.hip_processors(lens=lens).items()}
    torch.manual_seed(5)
    params = {t: {k: v.detach() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
    params["compressor"]["z_alpha_pre"] = torch.full_like(params["compressor"]["z_alpha_pre"], 4.5)   # a = 0.989: H ~ 2500 < 4095
    x = torch.randn(B, 32, 2, L) * 0.3
    with torch.no_grad():
        y = render_grafx(procs, x.cuda(), {t: {k: v.cuda() for k, v in d.items()} for t, d in params.items()}, rd_dev,
                         parameters_grad=False)[0].cpu()
    oprocs = {"eq": oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=lens["fsm_fir_len"]),
              "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=lens["iir_len"]),
              "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=lens["ir_len"])}
    with torch.no_grad():
        ref = render_grafx(oprocs, x, params, rd, parameters_grad=False)[0]
        ref64 = render_grafx(oprocs, x.double(), {t: {k: v.double() for k, v in d.items()} for t, d in params.items()}, rd,
                             parameters_grad=False)[0].float()
    assert_parity(y, ref, ref64, 1e-5, "console with long compressor poles")


def test_long_pole_rows_at_the_console_size_repeat_property():
    """9216-row scale is too long for the CPU oracle: 256 different rows against the oracle, then the same rows tiled to the
    stage's size must reproduce them bit for bit (rows are independent: a scheduling-order bug would break equality)."""
    from grafx_amd import ops

    R0, reps, L = 32, 64, 131072
    torch.manual_seed(11)
    x0 = torch.randn(R0, 2, L) * 0.4
    z0 = torch.linspace(4.0, 6.3, R0)[:, None]
    p0 = _params(R0, z0, 12)
    x = x0.cuda().repeat(reps, 1, 1)
    args = [p0[k].cuda().repeat(reps, 1) for k in ("log_threshold", "log_ratio", "log_knee", "z_alpha_pre")]
    y = ops.dynamics_fused(x, *args, smoother=1, iir_len=16383, knee="quadratic", gate=False)
    first = y[:R0].cpu()
    o = oracle.OracleCompressor(iir_len=16383)
    assert_parity(first, o(x0, **p0), o(x0.double(), **{k: v.double() for k, v in p0.items()}).float(), 1e-5,
                  "long-pole rows, console length")
    y = y.view(reps, R0, 2, L)
    assert torch.equal(y, y[:1].expand_as(y))
