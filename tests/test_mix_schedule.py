"""Host side of the fused routing sum (gfx_dynamics_fused_mix_f32): the per-row schedule that assigns every mix
destination one of four accumulators between its first and last source (grafx_amd.ops.mix_schedule)."""


def test_mix_schedule_colours_live_ranges():
    from grafx_amd import ops

    # the console: four buses of eight strips take turns in one accumulator, the send bus holds the other
    dests = [list(range(8 * k, 8 * k + 8)) for k in range(4)] + [list(range(32))]
    codes, n_acc, pre, post = ops.mix_schedule(dests, 32)
    assert pre == [] and post == []
    assert n_acc == 2
    assert all(c & 3 == 3 for c in codes)
    assert [(c >> 8) & 255 for c in codes if (c >> 8) & 255] + [(c >> 16) & 255 for c in codes if (c >> 16) & 255] in (
        [1, 2, 3, 4, 5], [5, 1, 2, 3, 4])
    assert ops.mix_schedule([[0, 1], []], 4) is None               # a destination without sources
    assert ops.mix_schedule([[1, 0]], 4) is None                   # not increasing
    assert ops.mix_schedule([[0, 3]] * 5, 4) is None               # five live at once
    assert ops.mix_schedule([[0, 3]] * 4, 4)[1] == 4
    # the master sum of the console: four bus compressors (the stage's rows 0..3) and the reverb return behind them
    codes, n_acc, pre, post = ops.mix_schedule([[0, 1, 2, 3, 4]], 4)
    assert n_acc == 1 and pre == [] and codes == [1, 1, 1, 1] and post == [(4, 1 | (1 << 8))]
    # a finished row in front of the stage's rows, and a destination made of extras alone is not this stage's business
    codes, n_acc, pre, post = ops.mix_schedule([[-3, 0, 1], [1, 5]], 2)
    assert pre == [(-3, 1)] and codes == [1, 1 | (1 << 8) | 2] and post == [(5, 2 | (2 << 16))] and n_acc == 2
    assert ops.mix_schedule([[-1, 7]], 4) is None


def test_overlap_of_buffer_views_is_exact_for_node_ranges():
    """ops._overlap (guards the multi-pass kernels against in-place output): node ranges of one (B, V, C, L) buffer
    interleave in memory, so their bounding address ranges intersect while no element is shared."""
    import torch

    from grafx_amd.ops import _overlap

    buf = torch.zeros(3, 10, 2, 64)
    assert not _overlap(buf[:, 2:5], buf[:, 5:8])
    assert _overlap(buf[:, 2:5], buf[:, 4:7])
    assert _overlap(buf[:, 2:5], buf[:, 2:5])
    assert not _overlap(buf[:, 0:1], buf[:, 9:10])
    assert not _overlap(buf[:, 2:5, :, :32], buf[:, 2:5, :, 32:])
    assert _overlap(buf[:, 2:5, :, :33], buf[:, 2:5, :, 32:])
    x = torch.zeros(4, 2, 64)
    assert _overlap(x, x) and _overlap(x[:3], x[1:]) and not _overlap(x[:2], x[2:])
    assert not _overlap(x, torch.zeros(4, 2, 64))


def test_block_structure_of_a_routing_sums_adjoint():
    """_block_fan on hand-made plans: blocks need contiguous sources with identical destination sets."""
    from types import SimpleNamespace

    import torch

    from grafx_amd.render.graph import _block_fan

    dev = torch.device("cpu")

    def plan(dests_per_slot):
        src, seg = [], [0]
        for rows in dests_per_slot:
            src += rows
            seg.append(len(src))
        return torch.tensor(src), torch.tensor(seg), len(dests_per_slot), None

    # console: four buses of eight strips (rows 64..95) + a send that takes them all
    console = plan([list(range(64 + 8 * k, 72 + 8 * k)) for k in range(4)] + [list(range(64, 96))])
    u0, k, m, idx, ptr = _block_fan(SimpleNamespace(), console, dev)
    assert (u0, k, m) == (64, 4, 8) and idx.tolist() == [0, 4, 1, 4, 2, 4, 3, 4] and ptr.tolist() == [0, 2, 4, 6, 8]
    # one destination fed by every source: a single block
    assert _block_fan(SimpleNamespace(), plan([[3, 4, 5, 6]]), dev)[:3] == (3, 1, 4)
    # every source on its own destination, gaps between the sources, unequal blocks: nothing to share
    assert _block_fan(SimpleNamespace(), plan([[0], [1], [2]]), dev) is None
    assert _block_fan(SimpleNamespace(), plan([[0, 1], [3, 4]]), dev) is None
    assert _block_fan(SimpleNamespace(), plan([[0, 1, 2], [3, 4]]), dev) is None
    # blocks of two inside groups of four
    assert _block_fan(SimpleNamespace(), plan([[0, 1, 2, 3], [0, 1], [2, 3]]), dev)[:3] == (0, 2, 2)
