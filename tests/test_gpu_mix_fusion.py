"""The routing sum fused into the dynamics kernel (gfx_dynamics_fused_mix_f32): a "mix" stage fed by a compressor / gate
stage alone (render/core.py:36-112 after dynamics.py:361-489) is produced by the compressor's one-shot tiles -- every
row's output AND the destination sums, bit for bit what the two separate kernels give."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(n, seed, slow_rows=()):
    g = torch.Generator().manual_seed(seed)
    p = {k: torch.randn(n, 1, generator=g) for k in ("log_threshold", "log_ratio", "log_knee")}
    z = torch.randn(n, 1, generator=g)
    for j in slow_rows:
        z[j] = 9.0            # pole 0.9999: history beyond the one-shot budget, the row kernel's row
    p["z_alpha"] = z
    return {k: v.cuda() for k, v in p.items()}


def _random_routing(n, J, seed):
    """J destinations over n rows with at most four live at a time (what the kernel's accumulators hold)."""
    import random

    from grafx_amd import ops

    rnd = random.Random(seed)
    for _ in range(1000):
        dests = []
        for d in range(J):
            lo = rnd.randrange(n)
            hi = rnd.randrange(lo, n)
            rows = sorted({lo, hi} | {j for j in range(lo, hi + 1) if rnd.random() < 0.5})
            dests.append(rows)
        sched = ops.mix_schedule(dests, n)
        if sched is not None:
            return dests, sched
    raise AssertionError("no schedule found")


@pytest.mark.parametrize("B,n,C,L,J,slow", [(3, 8, 2, 4096, 3, ()), (2, 32, 2, 20000, 5, (3, 17)), (4, 5, 1, 3001, 8, (0,)),
                                            (1, 1, 2, 1024, 1, ()), (2, 6, 2, 8191, 2, (5,)), (2, 7, 1, 2048, 4, ()),
                                            (2, 40, 2, 6000, 13, (1, 2, 39))])
@pytest.mark.parametrize("knee,gate", [("hard", False), ("quadratic", False), ("exponential", False), ("quadratic", True),
                                       ("hard", True)])
def test_fused_mix_equals_the_two_kernels(B, n, C, L, J, slow, knee, gate):
    from grafx_amd import ops

    torch.manual_seed(B * 100 + n + L)
    buf = torch.randn(B, n + n + J + 2, C, L, device="cuda")       # [inputs | outputs | mix | spare]
    x, y, mo = buf[:, :n], buf[:, n : 2 * n], buf[:, 2 * n : 2 * n + J]
    spare = buf[:, 2 * n + J :].clone()
    p = _params(n, L + J, slow)
    dests, (codes, n_acc, _, _) = _random_routing(n, J, seed=J * 1000 + n)
    kw = dict(smoother=1, iir_len=8193, knee=knee, gate=gate, param_rows=n)
    a = (p["log_threshold"], p["log_ratio"], p["log_knee"] if knee != "hard" else None, p["z_alpha"])
    want_y = ops.dynamics_fused(x, *a, **kw)                       # (B * n, C, L)
    want_y = want_y.view(B, n, C, L)
    want_m = torch.zeros(B, J, C, L, device="cuda")
    for d, rows in enumerate(dests):                               # sequential fp32 additions from 0.0, increasing row
        for j in rows:
            want_m[:, d] = want_m[:, d] + want_y[:, j]
    mix = {"sched": torch.tensor(codes, device="cuda"), "n_acc": n_acc, "out": mo}
    got_y = ops.dynamics_fused(x, *a, **kw, out=y, mix=mix)
    assert torch.equal(got_y, want_y)
    if L % 4:                                                      # the fused kernel takes whole aligned float4 only
        assert "done" not in mix
        return
    assert mix.get("done") is True
    assert torch.equal(mo, want_m)
    assert torch.equal(buf[:, 2 * n + J :], spare)                 # nothing written past the destinations


@pytest.mark.parametrize("C", [1, 2])
def test_fused_mix_with_rows_of_other_stages(C):
    """Destinations that also sum finished rows of the buffer in front of and behind the stage's rows (the console's master
    sum: four bus compressors and the reverb return): added in increasing row order, bit for bit the gather-sum."""
    from grafx_amd import ops

    torch.manual_seed(7 + C)
    B, n, L = 3, 4, 6000
    # buffer rows: [0,2) earlier stages | [2,6) inputs | [6,10) this stage's outputs | [10,12) a later-indexed finished stage
    #              | [12,15) mix destinations
    buf = torch.randn(B, 16, C, L, device="cuda")
    x, y, mo = buf[:, 2:6], buf[:, 6:10], buf[:, 12:15]
    p = _params(n, 3, slow_rows=(2,))
    dests_global = [[0, 6, 7, 11], [8, 9, 10], [1, 7, 9]]          # buffer rows per destination, increasing
    sched = ops.mix_schedule([[r - 6 for r in rows] for rows in dests_global], n)
    assert sched is not None
    codes, n_acc, pre, post = sched
    extras = torch.tensor([(6 + r - 12, c) for r, c in pre + post], device="cuda")
    kw = dict(smoother=1, iir_len=1023, knee="quadratic", gate=False, param_rows=n)
    a = (p["log_threshold"], p["log_ratio"], p["log_knee"], p["z_alpha"])
    want_y = ops.dynamics_fused(x, *a, **kw).view(B, n, C, L)
    ref = buf.clone()
    ref[:, 6:10] = want_y
    want_m = torch.zeros(B, 3, C, L, device="cuda")
    for d, rows in enumerate(dests_global):
        for r in rows:
            want_m[:, d] = want_m[:, d] + ref[:, r]
    mix = {"sched": torch.tensor(codes, device="cuda"), "n_acc": n_acc, "out": mo, "extras": extras, "n_pre": len(pre)}
    ops.dynamics_fused(x, *a, **kw, out=y, mix=mix)
    assert mix.get("done") is True
    assert torch.equal(buf[:, 6:10], want_y) and torch.equal(mo, want_m)
    assert torch.equal(buf[:, :6], ref[:, :6]) and torch.equal(buf[:, 10:12], ref[:, 10:12]) and torch.equal(buf[:, 15:], ref[:, 15:])


def test_fused_mix_declines_what_it_cannot_do():
    from grafx_amd import ops

    x = torch.randn(2, 4, 2, 2048, device="cuda")
    p = _params(4, 1)
    a = (p["log_threshold"], p["log_ratio"], None, p["z_alpha"])
    codes, n_acc, _, _ = ops.mix_schedule([[0, 1], [2, 3]], 4)
    for kw in (dict(smoother=0, iir_len=1), dict(smoother=1, iir_len=4097, schedule="rows")):
        mix = {"sched": torch.tensor(codes, device="cuda"), "n_acc": n_acc, "out": torch.empty(2, 2, 2, 2048, device="cuda")}
        ops.dynamics_fused(x, *a, knee="hard", gate=False, param_rows=4, mix=mix, **kw)
        assert "done" not in mix


@pytest.mark.parametrize("train", [False, True])
def test_console_render_with_and_without_the_fused_mix(train):
    """render_grafx on the bench's console graph: identical buffers with the fusion on and off (forward), identical
    gradients in training (the backward does not depend on how the forward summed)."""
    import bench
    from grafx_amd import ops
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    dev = torch.device("cuda")
    G = bench.console_graph(n_ch=8, n_bus=2)
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
    procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
    torch.manual_seed(3)
    B, L = 2, 32768
    x = torch.randn(B, 8, 2, L, device=dev)
    params = {t: {k: v.detach().to(dev) for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
    outs, seen = {}, {}
    real = ops.lib().gfx_dynamics_fused_mix_f32
    for flag in (True, False):
        ops.MIX_FUSION = flag
        try:
            if train:
                leaves = [v.requires_grad_(True) for d in params.values() for v in d.values()]
                for t in leaves:
                    t.grad = None
                out, _, buf = render_grafx(procs, x, params, rd)
                out.square().mean().backward()
                outs[flag] = (buf.detach().clone(), out.detach().clone(), [t.grad.clone() for t in leaves])
            else:
                with torch.no_grad():
                    out, _, buf = render_grafx(procs, x, params, rd, parameters_grad=False)
                outs[flag] = (buf.clone(), out.clone(), [])
        finally:
            ops.MIX_FUSION = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
    for a, b in zip(outs[True][2], outs[False][2]):
        # the same backward kernels on the same buffers, and no kernel of the training path adds with float atomics
        # (the compressor backward's per-row sums go through ordered partials): the same bits
        assert torch.equal(a, b)


def test_the_console_graph_takes_the_fused_path(monkeypatch):
    """The schedule of the bench graph has compressor -> mix (channel strips -> the buses and the send) and compressor ->
    reverb -> out (the master sum of the bus compressors and the reverb return; the reverb does not depend on the bus
    compressors and runs first): both routing sums are produced by the dynamics kernels, no gather-sum launch is left."""
    import bench
    from grafx_amd import ops
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    dev = torch.device("cuda")
    G = bench.console_graph(n_ch=8, n_bus=2)
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
    procs = {k: v.to(dev) for k, v in bench.hip_processors().items()}
    params = {t: {k: v.detach().to(dev) for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
    x = torch.randn(2, 8, 2, 16384, device=dev)
    calls = {"fanout": 0, "plain": 0}
    f0, g0 = ops.gather_sum_fanout, ops.gather_sum
    monkeypatch.setattr(ops, "gather_sum_fanout", lambda *a, **k: (calls.__setitem__("fanout", calls["fanout"] + 1), f0(*a, **k))[1])
    monkeypatch.setattr(ops, "gather_sum", lambda *a, **k: (calls.__setitem__("plain", calls["plain"] + 1), g0(*a, **k))[1])
    with torch.no_grad():
        render_grafx(procs, x, params, rd, parameters_grad=False)
    fused = dict(calls)
    ops.MIX_FUSION = False
    try:
        with torch.no_grad():
            render_grafx(procs, x, params, rd, parameters_grad=False)
    finally:
        ops.MIX_FUSION = True
    separate = {k: calls[k] - fused[k] for k in calls}
    # both sums -- strips -> buses + send, and bus compressors + reverb return -> out -- ride on compressor kernels
    assert sum(separate.values()) == 2 and sum(fused.values()) == 0, (fused, separate)


_FUSED_SEEN = []


@pytest.mark.parametrize("seed", range(10))
def test_random_graphs_render_identically_with_and_without_the_fused_mix(seed, monkeypatch):
    """Random DAGs (compressors, equalisers, reverbs and mix nodes wired at random, both schedulers, 3-D and 4-D inputs):
    the signal buffer is bit-identical whether the routing sums ride on the compressor kernel or run on their own; the
    last seed also checks that the fused path was taken by some of the graphs."""
    import random

    import grafx_amd.processors as P
    from grafx_amd import ops
    from grafx_amd.data import GRAFX, NodeConfigs, convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    rng = random.Random(500 + seed)
    torch.manual_seed(seed)
    G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
    n_src = rng.randint(2, 4)
    nodes = [G.add("in") for _ in range(n_src)]
    comps = []
    for s in list(nodes):                         # a compressor per source, so that mixes over compressor outputs exist
        c = G.add("compressor")
        G.connect(s, c)
        comps.append(c)
    nodes += comps
    for _ in range(rng.randint(2, 6)):
        kind = rng.choice(["eq", "compressor", "mix", "mix", "reverb"])
        v = G.add(kind)
        pool = comps if (kind == "mix" and rng.random() < 0.7) else nodes
        for s in rng.sample(pool, 1 if kind != "mix" else min(len(pool), rng.randint(2, 4))):
            G.connect(s, v)
        nodes.append(v)
    out = G.add("out")
    for s in rng.sample(nodes[n_src:], min(3, len(nodes) - n_src)):
        G.connect(s, out)
    L = rng.choice([1024, 2048, 4100, 6000])
    procs = {"eq": P.ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=257).cuda(),
             "compressor": P.Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False).cuda(),
             "reverb": P.STFTMaskedNoiseReverb(ir_len=1501, flashfftconv=False).cuda()}
    method = rng.choice(["beam", "greedy"])
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method=method)).to("cuda")
    params = {t: {k: v.cuda() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.3).items()}
    x = torch.randn(*((rng.randint(1, 3),) if rng.random() < 0.7 else ()), n_src, 2, L, device="cuda")
    real = ops.dynamics_fused

    def spy(*a, **k):
        out = real(*a, **k)
        if k.get("mix") is not None and k["mix"].get("done"):
            _FUSED_SEEN.append(seed)
        return out

    monkeypatch.setattr(ops, "dynamics_fused", spy)
    bufs = {}
    for flag in (True, False):
        ops.MIX_FUSION = flag
        try:
            with torch.no_grad():
                y, _, buf = render_grafx(procs, x, params, rd)
            bufs[flag] = (buf.clone(), y.clone())
        finally:
            ops.MIX_FUSION = True
    assert torch.equal(bufs[True][0], bufs[False][0]) and torch.equal(bufs[True][1], bufs[False][1])
    if seed == 9:
        assert len(set(_FUSED_SEEN)) >= 2, _FUSED_SEEN


@pytest.mark.parametrize("Cin", [1, 2])
def test_stereo_gain_with_the_fused_mix(Cin):
    """gfx_stereo_gain_mix_f32 (stereo.py:25-48 + render/core.py:36-112): the scaled rows and the routing sums, incl. rows of
    other stages, bit for bit what gfx_stereo_gain_f32 and the gather-sum give."""
    from grafx_amd import ops

    torch.manual_seed(21 + Cin)
    B, n, L = 3, 6, 5000
    xin = torch.randn(B, n, Cin, L, device="cuda")
    buf = torch.randn(B, 16, 2, L, device="cuda")                  # rows [2,8): this stage | 0, 1, 9: finished | [12,15): mix
    y, mo = buf[:, 2:8], buf[:, 12:15]
    lg = torch.randn(B * n, 2, device="cuda")
    dests_global = [[0, 2, 3, 7, 9], [4, 5, 6], [1, 3, 5]]
    codes, n_acc, pre, post = ops.mix_schedule([[r - 2 for r in rows] for rows in dests_global], n)
    extras = torch.tensor([(2 + r - 12, c) for r, c in pre + post], device="cuda")
    want_y = ops.stereo_gain(xin, lg).view(B, n, 2, L)
    ref = buf.clone()
    ref[:, 2:8] = want_y
    want_m = torch.zeros(B, 3, 2, L, device="cuda")
    for d, rows in enumerate(dests_global):
        for r in rows:
            want_m[:, d] = want_m[:, d] + ref[:, r]
    mix = {"sched": torch.tensor(codes, device="cuda"), "n_acc": n_acc, "out": mo, "extras": extras, "n_pre": len(pre)}
    ops.stereo_gain(xin, lg, out=y, mix=mix)
    assert mix.get("done") is True
    assert torch.equal(buf[:, 2:8], want_y) and torch.equal(mo, want_m)
    keep = [0, 1, 8, 9, 10, 11, 15]
    assert torch.equal(buf[:, keep], ref[:, keep])


def test_gain_stage_in_front_of_buses_renders_identically():
    """in -> eq -> gain -> {bus mix, send mix} -> out: the gain stage produces the bus sums (identical buffer, fewer launches)."""
    import grafx_amd.processors as P
    from grafx_amd import ops
    from grafx_amd.data import GRAFX, NodeConfigs, convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    torch.manual_seed(4)
    G = GRAFX(config=NodeConfigs(["eq", "gain"]))
    out = G.add("out")
    buses = [G.add("mix") for _ in range(2)]
    for ch in range(6):
        _, last = G.add_serial_chain(["in", "eq", "gain"])
        G.connect(last, buses[ch // 3])
        G.connect(last, out)
    for b in buses:
        G.connect(b, out)
    procs = {"eq": P.ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=257).cuda(), "gain": P.StereoGain().cuda()}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")
    params = {t: {k: v.cuda() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.3).items()}
    x = torch.randn(2, 6, 2, 4096, device="cuda")
    seen = []
    real = ops.stereo_gain

    def spy(*a, **k):
        o = real(*a, **k)
        seen.append(bool(k.get("mix") is not None and k["mix"].get("done")))
        return o

    bufs = {}
    for flag in (True, False):
        ops.MIX_FUSION = flag
        ops.stereo_gain = spy
        try:
            with torch.no_grad():
                yy, _, buf = render_grafx(procs, x, params, rd)
            bufs[flag] = (buf.clone(), yy.clone())
        finally:
            ops.MIX_FUSION, ops.stereo_gain = True, real
    assert torch.equal(bufs[True][0], bufs[False][0]) and torch.equal(bufs[True][1], bufs[False][1])
    assert seen[0] is True and seen[-1] is False


def test_fused_sum_of_a_source_row_waits_for_the_side_stream_copy():
    """in -> compressor -> mix(in, compressor) -> out (parallel dry / wet routing): the first stage reads the sources
    from the caller's tensor, so nothing before the fused kernel has joined the side stream that copies them into the
    buffer -- the kernel's `extras` read of the source row must.  Large rows, so that the copy is still running when
    the compressor kernel starts; many repetitions with a buffer whose previous contents are poison."""
    import grafx_amd.processors as P
    from grafx_amd import ops
    from grafx_amd.data import GRAFX, NodeConfigs, convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    torch.manual_seed(11)
    G = GRAFX(config=NodeConfigs(["compressor"]))
    src = [G.add("in") for _ in range(3)]
    out = G.add("out")
    for s in src:
        c, m = G.add("compressor"), G.add("mix")
        G.connect(s, c)
        G.connect(s, m)
        G.connect(c, m)
        G.connect(m, out)
    procs = {"compressor": P.Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False).cuda()}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to("cuda")
    params = {t: {k: v.cuda() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.3).items()}
    B, L = 48, 1 << 20                                       # 1.2 GB of sources: the copy takes ~0.4 ms
    x = torch.randn(B, 3, 2, L, device="cuda")
    seen = []
    real = ops.dynamics_fused

    def spy(*a, **k):
        o = real(*a, **k)
        seen.append(bool(k.get("mix") is not None and k["mix"].get("done")))
        return o

    ops.MIX_FUSION = False
    try:
        with torch.no_grad():
            want = render_grafx(procs, x, params, rd)[0].clone()
    finally:
        ops.MIX_FUSION = True
    ops.dynamics_fused = spy
    try:
        for rep in range(6):
            # poison the block the allocator will hand out for the next signal buffer
            poison = torch.full((B, rd.num_nodes, 2, L), float("nan"), device="cuda")
            del poison
            with torch.no_grad():
                got = render_grafx(procs, x, params, rd)[0]
            assert torch.equal(got, want), f"repetition {rep}: the fused sum read source rows before they were copied"
    finally:
        ops.dynamics_fused = real
    assert any(seen), "the fused routing sum was not taken: the test does not exercise the race"
