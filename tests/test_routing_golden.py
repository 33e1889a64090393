"""Host logic (graph -> tensors -> schedule -> RenderData -> render loop) against
golden routing data dumped from the reference (tests/golden/g8_*).  Bit-exact."""
import json
import os

import pytest
import torch

import oracle
from conftest import GOLDEN, assert_close
from grafx_amd.data import GRAFX, NodeConfigs, batch_grafx, convert_to_tensor
from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
from grafx_amd.utils import create_empty_parameters

with open(os.path.join(GOLDEN, "g8_routing.json")) as f:
    ROUTING = json.load(f)


def build_console(n_ch=32, n_bus=4):
    G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
    out_id = G.add("out")
    buses = [G.add("mix") for _ in range(n_bus)]
    send = G.add("mix")
    for ch in range(n_ch):
        _, last = G.add_serial_chain(["in", "eq", "compressor"])
        G.connect(last, buses[ch // (n_ch // n_bus)])
        G.connect(last, send)
    for b in buses:
        e, c = G.add("eq"), G.add("compressor")
        G.connect(b, e)
        G.connect(e, c)
        G.connect(c, out_id)
    r = G.add("reverb")
    G.connect(send, r)
    G.connect(r, out_id)
    return G


def build_cfg1():
    G = GRAFX(config=NodeConfigs(["gain", "biquad"]))
    _, last = G.add_serial_chain(["in", "gain", "biquad"])
    G.connect(last, G.add("out"))
    return G


def access_json(a):
    idx = a.idx.tolist() if isinstance(a.idx, torch.Tensor) else (list(a.idx) if a.idx is not None else None)
    return {"method": a.method, "idx": idx}


def render_json(rd):
    return {
        "method": rd.method, "num_nodes": int(rd.num_nodes), "max_order": int(rd.max_order),
        "siso_only": bool(rd.siso_only),
        "iter_list": [{
            "node_type": it.node_type,
            "source_reads": [access_json(a) for a in it.source_reads],
            "aggregations": [access_json(a) for a in it.aggregations],
            "parameter_read": access_json(it.parameter_read),
            "dest_write": access_json(it.dest_write),
        } for it in rd.iter_list],
    }


GRAPHS = {"cfg1": build_cfg1, "console32": build_console, "console8": lambda: build_console(8, 2)}


@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("method", ["beam", "greedy", "one-by-one"])
def test_schedule_and_render_data_bit_exact(name, method):
    G = GRAPHS[name]()
    if name != "cfg1":
        raw = ROUTING[f"{name}_raw"]
        G_raw = convert_to_tensor(G)
        assert G_raw.node_types.tolist() == raw["node_types"]
        assert G_raw.edge_indices.tolist() == raw["edge_indices"]
    G_t = reorder_for_fast_render(convert_to_tensor(G), method=method)
    want = ROUTING[f"{name}_{method}"]
    assert G_t.node_types.tolist() == want["tensor"]["node_types"]
    assert G_t.edge_indices.tolist() == want["tensor"]["edge_indices"]
    assert G_t.rendering_orders.tolist() == want["tensor"]["rendering_orders"]
    assert list(G_t.type_sequence) == want["tensor"]["type_sequence"]
    assert render_json(prepare_render(G_t)) == want["render"]


def test_console32_shape():
    G = build_console()
    assert G.number_of_nodes() == 111 and G.number_of_edges() == 142
    G_t = reorder_for_fast_render(convert_to_tensor(G), method="beam")
    assert G_t.type_sequence == ["in", "eq", "compressor", "mix", "eq", "compressor", "reverb", "out"]


def test_graph_order_on_grafx_object():
    G = reorder_for_fast_render(build_console(8, 2), method="beam")
    G_t = convert_to_tensor(G)
    want = ROUTING["console8_beam"]["tensor"]
    assert G_t.node_types.tolist() == want["node_types"]
    assert G_t.rendering_orders.tolist() == want["rendering_orders"]


def test_cfg1_render_with_oracle_processors(golden):
    g = golden("g8_render")
    G = build_cfg1()
    procs = {"gain": oracle.OracleStereoGain(), "biquad": oracle.OracleBiquadFilter(num_filters=1, fsm_fir_len=257)}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = {"gain": {"log_gain": g["cfg1_p_gain_log_gain"]},
              "biquad": {k: g[f"cfg1_p_biquad_{k}"] for k in ("Bs", "A1_pre", "A2_pre")}}
    y, _, buf = render_grafx(procs, g["cfg1_x"], params, rd)
    assert_close(y, g["cfg1_y"], 2e-6, "cfg1 y")
    assert_close(buf, g["cfg1_buf"], 2e-6, "cfg1 buffer")


def test_console8_batched_render_with_oracle_processors(golden):
    g = golden("g8_render")
    G = build_console(8, 2)
    procs = {
        "eq": oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=257),
        "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=255),
        "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=1501),
    }
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = {t: {k: g[f"console8_p_{t}_{k}"] for k in procs[t].parameter_size()} for t in procs}
    y, _, buf = render_grafx(procs, g["console8_x"], params, rd)
    assert_close(y, g["console8_y"], 5e-6, "console8 y")
    assert_close(buf[:, -8:], g["console8_buf_last8"], 5e-6, "console8 buffer tail")


def test_create_empty_parameters_shapes():
    G = build_console(8, 2)
    procs = {"eq": oracle.OracleParametricEqualizer(num_filters=6), "compressor": oracle.OracleCompressor(),
             "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=1501)}
    torch.manual_seed(0)
    p = create_empty_parameters(procs, G, std=0.5)
    assert p["eq"]["w0"].shape == (10, 1, 6)
    assert p["compressor"]["log_knee"].shape == (10, 1)
    assert p["reverb"]["init_log_magnitude"].shape == (1, 2, 193)


def test_graph_validation_and_batching():
    cfg = NodeConfigs(["eq"])
    G = GRAFX(config=cfg)
    a, b = G.add("in"), G.add("eq")
    G.connect(a, b)
    with pytest.raises(Exception):
        G.add("nope")
    with pytest.raises(Exception):
        G.connect(a, a)
    with pytest.raises(Exception):
        G.connect(a, b)  # duplicate
    Gw = GRAFX(config=cfg, invalid_op="warn")
    with pytest.warns(UserWarning):
        Gw.add("nope")
    Gm = GRAFX(config=cfg, invalid_op="mute")
    assert Gm.add("nope") is None
    with pytest.raises(Exception):
        GRAFX(invalid_op="bad")

    def small():
        H = GRAFX(config=cfg)
        _, last = H.add_serial_chain(["in", "eq"])
        H.connect(last, H.add("out"))
        return H

    GB = batch_grafx([small(), small(), small()])
    assert GB.number_of_nodes() == 9 and GB.batch and GB.counter == [3, 6, 9]
    with pytest.raises(Exception):
        batch_grafx([GB])
    inc, out = G.remove(b)
    assert len(inc) == 1 and not G.consecutive_ids


def test_mimo_config_indexing():
    cfg = NodeConfigs({"xover": {"inlets": ["main"], "outlets": ["low", "high"]}})
    assert not cfg.siso_only and cfg.max_num_outlets == 2
    assert cfg.outlet_to_index["xover"] == {"low": 0, "high": 1}
    assert "xover" in str(cfg)


# ---- g14: random DAGs built through the reference's graph API (tests/golden/make_golden.py g14) --------------------
with open(os.path.join(GOLDEN, "g14_random_graphs.json")) as f:
    RANDOM_GRAPHS = json.load(f)


def _rebuild(recipe):
    G = GRAFX(config=NodeConfigs(["gain", "biquad"]))
    for op in recipe:
        if op[0] == "add":
            G.add(op[1])
        else:
            G.connect(op[1], op[2])
    return G


@pytest.mark.parametrize("gi", range(len(RANDOM_GRAPHS)))
def test_random_graphs_schedule_and_route_bit_exactly(gi):
    """The same add / connect calls as the reference made, then all three schedulers: tensors, type sequence and every
    read / aggregate / write descriptor of the RenderData must be identical."""
    entry = RANDOM_GRAPHS[gi]
    G = _rebuild(entry["recipe"])
    for method, want in entry["schedules"].items():
        G_t = reorder_for_fast_render(convert_to_tensor(G), method=method)
        assert G_t.node_types.tolist() == want["tensor"]["node_types"], method
        assert G_t.edge_indices.tolist() == want["tensor"]["edge_indices"], method
        assert G_t.rendering_orders.tolist() == want["tensor"]["rendering_orders"], method
        assert list(G_t.type_sequence) == want["tensor"]["type_sequence"], method
        assert render_json(prepare_render(G_t)) == want["render"], method


@pytest.mark.parametrize("gi", range(len(RANDOM_GRAPHS)))
def test_random_graphs_render_like_the_reference(golden, gi):
    """Our render loop (with the oracle's StereoGain / BiquadFilter) on the reference's inputs and parameters: the
    output and every node's signal in the buffer."""
    g = golden("g14_random_graphs")
    entry = RANDOM_GRAPHS[gi]
    G = _rebuild(entry["recipe"])
    procs = {"gain": oracle.OracleStereoGain(), "biquad": oracle.OracleBiquadFilter(num_filters=2, fsm_fir_len=65)}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = {t: {k: g[f"g{gi:02d}_p_{t}_{k}"] for k in ks} for t, ks in entry["params"].items()}
    y, _, buf = render_grafx(procs, g[f"g{gi:02d}_x"], params, rd)
    assert_close(y, g[f"g{gi:02d}_y"], 5e-6, "output")
    assert_close(buf, g[f"g{gi:02d}_buf"], 5e-6, "signal buffer")


# ---- g15: common_parameters (one row per graph node, handed to every processor) ----------------------------------
class TrimmedGain(torch.nn.Module):
    """The user-defined processor the fixture was rendered with: y = x * exp(log_gain) * trim (trim is the common one)."""

    def forward(self, input_signals, log_gain, trim):
        return input_signals * torch.exp(log_gain)[..., None] * trim[..., None]

    def parameter_size(self):
        return {"log_gain": 2}


with open(os.path.join(GOLDEN, "g15_common_parameters.json")) as f:
    COMMON_GRAPHS = json.load(f)


@pytest.mark.parametrize("gi", range(len(COMMON_GRAPHS)))
def test_common_parameters_render_like_the_reference(golden, gi):
    """render/graph.py:72-75, 132-141: common parameters are expanded over the batch and read by destination index;
    batched (4-D) and unbatched (3-D) inputs alternate in the fixture."""
    g = golden("g15_common_parameters")
    entry = COMMON_GRAPHS[gi]
    G = GRAFX(config=NodeConfigs(["gain", "trim"]))
    for op in entry["recipe"]:
        G.add(op[1]) if op[0] == "add" else G.connect(op[1], op[2])
    procs = {"gain": TrimmedGain(), "trim": TrimmedGain()}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    tag = f"g{gi:02d}"
    params = {t: {k: g[f"{tag}_p_{t}_{k}"] for k in ks} for t, ks in entry["params"].items()}
    y, _, buf = render_grafx(procs, g[f"{tag}_x"], params, rd, common_parameters={"trim": g[f"{tag}_trim"]})
    assert_close(y, g[f"{tag}_y"], 1e-6, "output")
    assert_close(buf, g[f"{tag}_buf"], 1e-6, "signal buffer")


def test_node_configs_public_helpers():
    """get_default_config / unpack_list / unpack_dict (reference data/configs.py:71-120)."""
    cfg = NodeConfigs(["eq"])
    assert cfg.get_default_config("in") == {"inlets": [], "outlets": ["main"]}
    assert cfg.get_default_config("out") == {"inlets": ["main"], "outlets": []}
    assert cfg.get_default_config("mix") == cfg.get_default_config("anything") == {"inlets": ["main"], "outlets": ["main"]}
    cfg.unpack_list(["in", "x", "out"])
    assert cfg.node_types == ["in", "x", "out"] and cfg.siso_only and cfg.num_inlets == {"in": 0, "x": 1, "out": 1}
    cfg.unpack_dict({"split": {"inlets": ["main"], "outlets": ["low", "high"]}})
    assert not cfg.siso_only and cfg.max_num_outlets == 2 and cfg.outlet_to_index["split"] == {"low": 0, "high": 1}


# ---- g16: the reference's autograd gradients through render_grafx on random graphs -------------------------------
with open(os.path.join(GOLDEN, "g16_render_gradients.json")) as f:
    GRADIENT_GRAPHS = json.load(f)


@pytest.mark.parametrize("gi", range(len(GRADIENT_GRAPHS)))
def test_render_gradients_with_oracle_processors_match_the_reference(golden, gi):
    """Our (generic, taped) render loop with the oracle processors against the reference's own gradients: every
    parameter, the input, and which parameters receive no gradient at all."""
    g = golden("g16_render_gradients")
    entry = GRADIENT_GRAPHS[gi]
    G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
    for op in entry["recipe"]:
        G.add(op[1]) if op[0] == "add" else G.connect(op[1], op[2])
    procs = {"eq": oracle.OracleParametricEqualizer(num_filters=4, fsm_fir_len=257),
             "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=255),
             "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=1501)}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    tag, L = f"g{gi:02d}", entry["L"]
    params = {t: {k: g[f"{tag}_p_{t}_{k}"].clone().requires_grad_(True) for k in ks} for t, ks in entry["params"].items()}
    x = g[f"{tag}_x"].clone().requires_grad_(True)
    y, _, buf = render_grafx(procs, x, params, rd, input_signal_grad=True)
    w = torch.linspace(0.5, 1.5, L)
    loss = (y * w).square().mean()
    if entry["buffer_loss"]:
        loss = loss + 0.3 * (buf[:, buf.shape[1] // 2] * w).abs().mean()
    loss.backward()
    assert_close(x.grad, g[f"{tag}_gx"], 1e-4, "input gradient")
    nograd = {tuple(k) for k in entry["nograd"]}
    for t, d in params.items():
        for k, v in d.items():
            assert (v.grad is None) == ((t, k) in nograd), (t, k)
            if v.grad is not None:
                want = g[f"{tag}_g_{t}_{k}"]
                assert (v.grad - want).abs().max() <= 1e-3 * want.abs().max().clamp_min(1e-8), (t, k)
