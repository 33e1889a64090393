"""Render-order entry points accepting GRAFX or GRAFXTensor (mirrors
grafx.render.order.graph — reference src/grafx/render/order/graph.py:15-94)."""
import networkx as nx

from ...data.conversion import convert_to_tensor
from ...data.graph import GRAFX
from ...data.tensor import GRAFXTensor
from .tensor import compute_render_order_tensor, node_id_from_render_order, return_render_ordered_tensor


def compute_render_order(G_any, method="beam", **kwargs):
    if isinstance(G_any, GRAFX):
        return compute_render_order_tensor(convert_to_tensor(G_any), method, **kwargs)
    if isinstance(G_any, GRAFXTensor):
        return compute_render_order_tensor(G_any, method, **kwargs)
    raise Exception(f"Invalid graph type: {type(G_any)}")


def reorder_for_fast_render(G_any, method="beam", **kwargs):
    if isinstance(G_any, GRAFX):
        return return_render_ordered_graph(G_any, method, **kwargs)
    if isinstance(G_any, GRAFXTensor):
        return return_render_ordered_tensor(G_any, method, **kwargs)
    raise Exception(f"Invalid input type: {type(G_any)}")


def return_render_ordered_graph(G, method, **kwargs):
    type_sequence, render_order = compute_render_order(G, method, **kwargs)
    for node, level in zip(G.nodes, render_order):
        G.nodes[node]["rendering_order"] = level.item()
    new_ids = node_id_from_render_order(render_order).tolist()
    G = nx.relabel_nodes(G, mapping=dict(enumerate(new_ids)))
    G = get_sorted_graph(G)
    G.type_sequence = [G.config.node_types[t] for t in type_sequence]
    G.rendering_order_method = method
    return G


def get_sorted_graph(G):
    H = GRAFX()
    H.add_nodes_from(sorted(G.nodes(data=True)))
    H.add_edges_from(sorted(G.edges(data=True)))
    H.graph = G.graph.copy()
    return H
