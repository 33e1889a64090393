"""GPU parity at BASELINE's full sizes (L = 131072, default-scale filter lengths).

The oracle renders a couple of graphs of the headline console in well under a second, so the
headline configuration itself (111 nodes, fsm_fir_len 4001 / iir_len 16383 / ir_len 60001) is compared
sample by sample at batch 2; per-processor checks cover cfg2 / cfg3 shapes on a few rows, both length
parities (reference-default even lengths exercise the odd-P aliasing path end to end)."""
import pytest
import torch

import oracle
from conftest import assert_close, assert_parity, rel_err, rel_err_rows
from test_routing_golden import build_console

pytestmark = pytest.mark.gpu
L = 131072


def _params(procs, G, std, seed):
    from grafx_amd.utils import create_empty_parameters

    torch.manual_seed(seed)
    return {t: {k: v.detach() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=std).items()}


@pytest.mark.parametrize("lens", [dict(fsm_fir_len=4001, iir_len=16383, ir_len=60001),
                                  dict(fsm_fir_len=4000, iir_len=16384, ir_len=60000)])
def test_headline_console_graph_batch2(lens):
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render

    G = build_console(32, 4)
    hip = {"eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=lens["fsm_fir_len"]).cuda(),
           "compressor": Compressor(energy_smoother="iir", iir_len=lens["iir_len"], flashfftconv=False).cuda(),
           "reverb": STFTMaskedNoiseReverb(ir_len=lens["ir_len"], flashfftconv=False).cuda()}
    cpu = {"eq": oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=lens["fsm_fir_len"]),
           "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=lens["iir_len"]),
           "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=lens["ir_len"])}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = _params(hip, G, 0.1, 7)
    torch.manual_seed(8)
    x = torch.randn(2, 32, 2, L)
    with torch.no_grad():
        want, _, wbuf = render_grafx(cpu, x, params, rd, parameters_grad=False)
        dev = {t: {k: v.cuda() for k, v in d.items()} for t, d in params.items()}
        got, _, gbuf = render_grafx(hip, x.cuda(), dev,
                                    prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda"))
    if lens["fsm_fir_len"] % 2 == 1:  # reference == linear convolution: plain tolerance
        assert_close(got.cpu(), want, 1e-5, "console output")
        # every intermediate node too (eq outputs, compressor outputs, buses, reverb)
        for lo, hi, name in [(32, 64, "eq"), (64, 96, "compressor"), (96, 101, "mix"), (109, 110, "reverb")]:
            assert_close(gbuf[:, lo:hi].cpu(), wbuf[:, lo:hi], 1e-5, name)
        return
    # reference-default (even) lengths: every convolve() inverts a ~135071-point spectrum on a 135070-point
    # grid in fp32; five such stages in series leave the reference's own fp32 render a few 1e-5
    # from a float64 evaluation, so the float64 tie-breaker applies (conftest.assert_parity)
    with torch.no_grad():
        p64 = {t: {k: v.double() for k, v in d.items()} for t, d in params.items()}
        cpu64 = {k: m.double() for k, m in cpu.items()}
        want64, _, _ = render_grafx(cpu64, x.double(), p64, rd, parameters_grad=False)
    assert_parity(got.cpu(), want, want64.float(), 2e-5, "console output (odd-P aliasing path)")


def test_cfg2_peq_rows_full_length():
    from grafx_amd.processors import ParametricEqualizer

    torch.manual_seed(2)
    Lc = 480000  # BASELINE configs[1]: 10 s @ 48 kHz
    # the four tracks at 1, 1e-2, 1e-4 and 1: with upstream's 4000 taps the rows of a chirp-z pair are different tracks, and
    # each must be right relative to ITSELF (the per-call peak would hide a quiet track's error behind its loud partner)
    x = torch.randn(4, 1, Lc) * torch.tensor([1.0, 1e-2, 1e-4, 1.0])[:, None, None]
    p = {k: torch.randn(4, 1, 6) for k in ("w0", "q_inv", "log_gain")}
    for N in (4001, 4000):
        m = ParametricEqualizer(num_filters=6, processor_channel="mono", flashfftconv=False, fsm_fir_len=N)
        with torch.no_grad():
            y = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        ref = oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=N)(x, **p)
        ref64 = oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=N)(x.double(), **{k: v.double() for k, v in p.items()})
        assert_parity(y, ref, ref64, 1e-5, f"cfg2 N={N}")
        ours, noise = rel_err_rows(y, ref64)[0], rel_err_rows(ref, ref64.float())[0]
        assert rel_err_rows(y, ref)[0] <= 1e-5 or ours <= max(1e-5, noise), \
            f"cfg2 N={N}: worst row vs float64 {ours:.2e} of its own peak (the reference's own: {noise:.2e})"


def test_cfg3_reverb_rows_full_length():
    from grafx_amd.processors import STFTMaskedNoiseReverb

    torch.manual_seed(3)
    Lc = 240000  # BASELINE configs[2]: 5 s stereo
    x = torch.randn(2, 2, Lc)
    p = {k: torch.randn(2, 2, 193) for k in ("init_log_magnitude", "delta_log_magnitude")}
    for ir_len in (60001, 60000):
        m = STFTMaskedNoiseReverb(ir_len=ir_len, flashfftconv=False)
        with torch.no_grad():
            y = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        ref = oracle.OracleSTFTMaskedNoiseReverb(ir_len=ir_len)(x, **p)
        assert_close(y, ref, 1e-5, f"cfg3 ir_len={ir_len}")


def test_compressor_extreme_poles_full_length():
    """One-pole coefficients from very fast to the clamp (a = 1 - 1e-5, where the a^N term matters)."""
    from grafx_amd.processors import Compressor

    torch.manual_seed(4)
    z = torch.tensor([[-6.0], [0.0], [4.0], [8.0], [20.0]])
    x = torch.randn(5, 2, L) * torch.tensor([1.0, 0.3, 0.1, 1.0, 0.5])[:, None, None]
    p = {"log_threshold": torch.randn(5, 1), "log_ratio": torch.randn(5, 1), "log_knee": torch.randn(5, 1), "z_alpha_pre": z}
    for iir_len in (16383, 16384):
        m = Compressor(energy_smoother="iir", iir_len=iir_len, flashfftconv=False)
        with torch.no_grad():
            y = m.cuda()(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
        ref = oracle.OracleCompressor(iir_len=iir_len)(x, **p)
        ref64 = oracle.OracleCompressor(iir_len=iir_len)(x.double(), **{k: v.double() for k, v in p.items()})
        assert_parity(y, ref, ref64, 1e-5, f"compressor iir_len={iir_len}")


@pytest.mark.gpu
@pytest.mark.parametrize("iir_len", [16383, 1023])
def test_time_chunked_dynamics_equals_the_serial_scan(iir_len):
    """With few rows gfx_dynamics_fused_f32 splits every row into time chunks that re-scan iir_len samples of
    history; with many rows it walks each row serially.  Same rows, both ways, including a pole clamped at
    1 - 1e-5 (a^N = 0.85: the truncation term matters) and a hard gate."""
    import torch

    from grafx_amd import ops

    torch.manual_seed(9)
    R, Lc = 4, 131072
    x = torch.randn(R, 2, Lc, device="cuda") * torch.linspace(0.05, 1.0, Lc, device="cuda")
    p = dict(log_threshold=torch.randn(R, 1, device="cuda") - 2, log_ratio=torch.randn(R, 1, device="cuda"),
             log_knee=torch.randn(R, 1, device="cuda"), z_alpha=torch.tensor([[20.0], [6.0], [2.0], [-1.0]], device="cuda"))
    for knee, gate in (("quadratic", False), ("hard", True)):
        few = ops.dynamics_fused(x, p["log_threshold"], p["log_ratio"], p["log_knee"], p["z_alpha"], smoother=1,
                                 iir_len=iir_len, knee=knee, gate=gate)
        reps = 512  # 2048 rows: serial walk
        many = ops.dynamics_fused(x.repeat(reps, 1, 1), *(p[k].repeat(reps, 1) for k in
                                                           ("log_threshold", "log_ratio", "log_knee", "z_alpha")),
                                  smoother=1, iir_len=iir_len, knee=knee, gate=gate)
        assert (few - many[:R]).abs().max() <= 1e-5 * many[:R].abs().max(), (knee, gate)


def test_headline_batch_256_repeats_the_checked_batch_of_2():
    """BASELINE configs[3] at its full size (256 graphs, a 30 GB signal buffer: row offsets beyond 2^32 bytes, 180 k
    workgroups per launch).  Graphs of a batch never interact, so a batch that repeats two graphs 128 times must
    reproduce -- in every one of its 256 slots, for every node -- what the two graphs give on their own (the batch-2
    render is the one compared with the oracle sample by sample above)."""
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render

    G = build_console(32, 4)
    hip = {"eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=4001).cuda(),
           "compressor": Compressor(energy_smoother="iir", iir_len=16383, flashfftconv=False).cuda(),
           "reverb": STFTMaskedNoiseReverb(ir_len=60001, flashfftconv=False).cuda()}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")
    params = {t: {k: v.cuda() for k, v in d.items()} for t, d in _params(hip, G, 0.1, 7).items()}
    torch.manual_seed(8)
    x2 = torch.randn(2, 32, 2, L, device="cuda")
    with torch.no_grad():
        y2, _, buf2 = render_grafx(hip, x2, params, rd)
        x = x2.repeat(128, 1, 1, 1)
        y, _, buf = render_grafx(hip, x, params, rd)
    del x
    assert y.shape == (256, 1, 2, L) and buf.shape == (256, 111, 2, L)
    peak = buf2.abs().amax(dim=(0, 2, 3))                      # per node
    for b in range(0, 256, 2):
        pair = buf[b : b + 2]
        # sources: bit for bit.  Everything behind the first equaliser: to rounding -- the batch of 2 runs its 4001-tap
        # convolutions on the one-tile-per-workgroup kernel, the batch of 256 on the persistent hand-scheduled one (same
        # tiles, different butterfly forms: radix-2 DIT with fused multiply-adds vs the compiler's radix-4 passes)
        assert torch.equal(pair[:, :32], buf2[:, :32]), f"graphs {b}, {b + 1}: a source row differs"
        assert ((pair[:, 32:] - buf2[:, 32:]).abs().amax(dim=(0, 2, 3)) <= 4e-6 * peak[32:]).all(), f"graphs {b}, {b + 1}"
    assert (y - y2.repeat(128, 1, 1, 1)).abs().max() <= 4e-6 * y2.abs().max()


def test_cfg2_full_batch_1024_rows_repeat_the_checked_rows():
    """BASELINE configs[1] at its full size: ParametricEqualizer(6, mono), 1024 x 1 x 480000, N = 4001 (30 tiles per
    row, 30720 workgroups, 1024 distinct filters).  Rows never interact, so a batch that repeats the four rows checked
    against the oracle above (same seed) 256 times must reproduce them in every slot (to rounding: see below)."""
    from grafx_amd.processors import ParametricEqualizer

    torch.manual_seed(2)
    Lc = 480000
    x4 = torch.randn(4, 1, Lc)
    p4 = {k: torch.randn(4, 1, 6) for k in ("w0", "q_inv", "log_gain")}
    m = ParametricEqualizer(num_filters=6, processor_channel="mono", flashfftconv=False, fsm_fir_len=4001).cuda()
    with torch.no_grad():
        y4 = m(x4.cuda(), **{k: v.cuda() for k, v in p4.items()})
        ref = oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=4001)(x4, **p4)
        ref64 = oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=4001)(x4.double(), **{k: v.double() for k, v in p4.items()})
        assert_parity(y4.cpu(), ref, ref64, 1e-5, "cfg2 rows")
        x = x4.cuda().repeat(256, 1, 1)
        y = m(x, **{k: v.cuda().repeat(256, 1, 1) for k, v in p4.items()})
    assert y.shape == (1024, 1, Lc)
    # (the 4-row call runs on the one-tile-per-workgroup kernel, the full batch on the persistent hand-scheduled one:
    # same tiles, different butterfly forms, so the twins agree to rounding, not bit for bit)
    d = (y.view(256, 4, 1, Lc) - y4.expand(256, 4, 1, Lc)).abs().amax(dim=(0, 2, 3))
    assert (d <= 2e-6 * y4.abs().amax(dim=(1, 2))).all(), f"a row of the full batch differs from its checked twin: {d}"


def test_cfg3_full_batch_512_rows_repeat_the_checked_rows():
    """BASELINE configs[2] at its full size: STFTMaskedNoiseReverb(ir_len=60001), 512 x 2 x 240000: 8 partitions, 37
    windows per row-channel, a 1.2 GB spectrum workspace.  Same repeat-the-checked-rows property, bit for bit: both
    calls run the same kernels, and the energy normalisation of the impulse response sums in a fixed order."""
    from grafx_amd.processors import STFTMaskedNoiseReverb

    torch.manual_seed(3)
    Lc = 240000
    x2 = torch.randn(2, 2, Lc)
    p2 = {k: torch.randn(2, 2, 193) for k in ("init_log_magnitude", "delta_log_magnitude")}
    m = STFTMaskedNoiseReverb(ir_len=60001, flashfftconv=False).cuda()
    with torch.no_grad():
        y2 = m(x2.cuda(), **{k: v.cuda() for k, v in p2.items()})
        assert_close(y2.cpu(), oracle.OracleSTFTMaskedNoiseReverb(ir_len=60001)(x2, **p2), 1e-5, "cfg3 rows")
        y = m(x2.cuda().repeat(256, 1, 1), **{k: v.cuda().repeat(256, 1, 1) for k, v in p2.items()})
    assert y.shape == (512, 2, Lc)
    assert torch.equal(y.view(256, 2, 2, Lc), y2.expand(256, 2, 2, Lc)), "a row of the full batch differs from its checked twin"


@pytest.mark.parametrize("iir_len", [16383, 300, 40])
def test_oneshot_schedule_of_the_fused_dynamics_equals_the_row_schedule(iir_len):
    """gfx_dynamics_fused_ws_f32 (dependency-free one-shot tiles that re-read their smoother history, chosen per row on
    the device) against the row-streaming kernel: poles from instant to the clamp at 1 - 1e-5 in ONE call, so that rows
    taken by the one-shot grid and rows left to the row kernel (long memory, or a live truncation term at short
    iir_len) sit side by side; ragged length, mono and stereo, shared parameter rows, and the kept scan `u1`."""
    from grafx_amd import ops

    torch.manual_seed(12)
    for C, Lc in ((2, 131072), (1, 5001), (2, 1024), (2, 1028)):
        n, B = 8, 3
        x = torch.randn(B, n, C, Lc, device="cuda") * torch.linspace(0.05, 1.0, Lc, device="cuda")
        p = dict(log_threshold=torch.randn(n, 1, device="cuda") - 2, log_ratio=torch.randn(n, 1, device="cuda"),
                 log_knee=torch.randn(n, 1, device="cuda"),
                 z_alpha=torch.tensor([[20.0], [9.0], [6.0], [2.3], [2.0], [0.0], [-3.0], [-12.0]], device="cuda"))
        for knee, gate in (("quadratic", False), ("hard", True)):
            kw = dict(smoother=1, iir_len=iir_len, knee=knee, gate=gate, param_rows=n)
            ua = torch.empty(B * n, Lc, device="cuda")
            ub = torch.full((B * n, Lc), float("nan"), device="cuda")
            a = ops.dynamics_fused(x, p["log_threshold"], p["log_ratio"], p["log_knee"], p["z_alpha"], schedule="rows", u1_out=ua, **kw)
            b = ops.dynamics_fused(x, p["log_threshold"], p["log_ratio"], p["log_knee"], p["z_alpha"], schedule="oneshot", u1_out=ub, **kw)
            c = ops.dynamics_fused(x, p["log_threshold"], p["log_ratio"], p["log_knee"], p["z_alpha"], schedule="oneshot", **kw)
            assert torch.isfinite(b).all() and torch.isfinite(ub).all()
            assert (a - b).abs().max() <= 5e-6 * a.abs().max(), (C, Lc, knee, gate)
            # (without `u1` the few long-memory rows run time-chunked: scans restarted per chunk round differently from a
            # whole-row scan when the pole sits at the clamp and the FIR is 40 taps short -- 2e-5 of the peak)
            assert (b - c).abs().max() <= 5e-5 * a.abs().max()
            assert (ua - ub).abs().max() <= 5e-6 * ua.abs().max(), (C, Lc, knee, gate)


def test_oneshot_dynamics_matches_the_oracle_on_a_loud_to_silent_signal():
    """The one-shot tiles drop smoother taps below 1e-12: a passage 120 dB down right after a loud one is where a
    dropped tail would show.  Compared with the CPU oracle (the reference's FFT convolution with the 16383-tap FIR)."""
    import oracle
    from grafx_amd.processors import Compressor

    torch.manual_seed(3)
    Lc = 40960
    x = torch.randn(6, 2, Lc)
    x[:, :, 20000:] *= 1e-6
    m = Compressor(energy_smoother="iir", iir_len=16383, flashfftconv=False).cuda()
    o = oracle.OracleCompressor(energy_smoother="iir", iir_len=16383)
    p = dict(log_threshold=torch.randn(6, 1) - 3, log_ratio=torch.randn(6, 1), log_knee=torch.randn(6, 1),
             z_alpha_pre=torch.tensor([[2.0], [1.0], [0.0], [-1.0], [-4.0], [3.0]]))
    with torch.no_grad():
        want = o(x.double(), **{k: v.double() for k, v in p.items()}).float()
        got = m(x.cuda(), **{k: v.cuda() for k, v in p.items()}).cpu()
    for sl in (slice(0, 20000), slice(20000, Lc)):
        assert (got[..., sl] - want[..., sl]).abs().max() <= 1e-5 * want[..., sl].abs().max()


def test_console_training_step_at_batch_32_matches_the_oracles_autograd():
    """One training step of the headline console (111 nodes, 4001 / 16383 / 60001 taps, L = 131072) at batch 32 -- the
    persistent convolution and correlation kernels, the one-shot compressor tiles forward and backward, the fused routing
    sums and their adjoint, the partitioned convolution's gradient, the stage-wise backward of render_grafx -- against torch
    autograd through the CPU oracle on the same inputs: output and every shared-parameter gradient.  (The oracle's tape
    for 32 graphs is ~22 GB of host memory and half a minute of CPU time.)"""
    import psutil

    from grafx_amd.data import convert_to_tensor
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    if psutil.virtual_memory().available < 72 * 2**30:
        pytest.skip("the oracle's autograd tapes for 32 graphs need ~22 GB (float32) and ~44 GB (float64) of host memory")
    B = 32
    G = build_console(32, 4)
    lens = dict(fsm_fir_len=4001, iir_len=16383, ir_len=60001)
    hip = {"eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=lens["fsm_fir_len"]).cuda(),
           "compressor": Compressor(energy_smoother="iir", iir_len=lens["iir_len"], flashfftconv=False).cuda(),
           "reverb": STFTMaskedNoiseReverb(ir_len=lens["ir_len"], flashfftconv=False).cuda()}
    cpu = {"eq": oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=lens["fsm_fir_len"]),
           "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=lens["iir_len"]),
           "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=lens["ir_len"])}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    torch.manual_seed(21)
    params = create_empty_parameters(hip, G, std=0.1)
    x = torch.randn(B, 32, 2, L)
    y_ref = render_grafx(cpu, x, params, rd)[0]
    y_ref.square().mean().backward()
    ref = [(n, p.grad.clone()) for n, p in params.named_parameters()]
    y_ref = y_ref.detach()
    for p in params.parameters():
        p.grad = None
    import copy

    params64 = copy.deepcopy(params).double()   # float64 tie-breaker: the same render differentiated in double
    render_grafx(cpu, x.double(), params64, rd)[0].square().mean().backward()
    ref64 = [p.grad.float() for p in params64.parameters()]
    dev = params.cuda()
    y = render_grafx(hip, x.cuda(), dev, rd.to("cuda"))[0]
    y.square().mean().backward()
    assert_close(y.detach().cpu(), y_ref, 1e-5, "console output (training forward, batch 32)")
    for (name, want), got, w64 in zip(ref, dev.parameters(), ref64):
        assert got.grad is not None, name
        assert_parity(got.grad.cpu(), want, w64, 1e-5, f"gradient of {name}")
