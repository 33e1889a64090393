"""GRAFX -> GRAFXTensor (mirrors grafx.data.conversion — reference src/grafx/data/conversion.py:8-85)."""
import networkx as nx
import torch

from .tensor import GRAFXTensor


def convert_to_tensor(G):
    cfg = G.config
    if not G.consecutive_ids:
        G = _relabel_consecutive(G)
    nodes = sorted(G.nodes(data=True))
    edges = sorted(G.edges(data=True))

    node_types = torch.tensor([cfg.node_type_to_index[d["node_type"]] for _, d in nodes], dtype=torch.long)
    rendering_orders = None
    if G.rendering_order_method is not None:
        rendering_orders = torch.tensor([d.get("rendering_order", -1) for _, d in nodes], dtype=torch.long)

    edge_indices = torch.stack([torch.tensor([s for s, _, _ in edges]), torch.tensor([d for _, d, _ in edges])])

    edge_types = None
    if not cfg.siso_only:
        pairs = []
        for s, d, data in edges:
            outlet = cfg.outlet_to_index[G.nodes[s]["node_type"]][data["outlet"]]
            inlet = cfg.inlet_to_index[G.nodes[d]["node_type"]][data["inlet"]]
            pairs.append([outlet, inlet])
        edge_types = torch.tensor(pairs)

    return GRAFXTensor(
        node_types=node_types,
        edge_indices=edge_indices,
        edge_types=edge_types,
        rendering_order_method=G.rendering_order_method,
        rendering_orders=rendering_orders,
        type_sequence=G.type_sequence,
        counter=G.counter,
        batch=G.batch,
        config=G.config,
        config_hash=G.config_hash,
        invalid_op=G.invalid_op,
    )


def _relabel_consecutive(G):
    mapping = {old: new for new, old in enumerate(G.nodes())}
    return nx.relabel_nodes(G, mapping, copy=True)
