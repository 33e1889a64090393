"""Graph / parameter utilities (mirrors grafx.utils — reference src/grafx/utils.py:8-174)."""
import torch
import torch.nn as nn

from .data.tensor import GRAFXTensor


def get_node_ids_from_type(G, node_type):
    return [n for n, d in G.nodes(data=True) if d["node_type"] == node_type]


def count_nodes_per_type(G, types_to_count=None):
    if types_to_count is not None:
        counts = {k: 0 for k in types_to_count}
    elif G.config is not None:
        counts = {k: 0 for k in G.config.node_types}
    else:
        counts = {}
    for _, d in G.nodes(data=True):
        t = d["node_type"]
        if types_to_count is None:
            counts[t] = counts.get(t, 0) + 1
        elif t in types_to_count:
            counts[t] += 1
    return counts


def create_empty_parameters(processors, G, std=1e-2):
    """``nn.ParameterDict`` {type: {name: std*randn(num_nodes, *shape)}} (utils.py:60-87)."""
    counts = count_nodes_per_type(G, processors)
    out = {}
    for t in processors:
        out[t] = create_empty_parameters_from_shape_dict(processors[t].parameter_size(), counts[t], std=std)
    return nn.ParameterDict(out)


def create_empty_parameters_from_shape_dict(parameter_shapes, num_nodes, std=1e-2, root=True, device="cpu"):
    if isinstance(parameter_shapes, dict):
        return nn.ParameterDict({
            k: create_empty_parameters_from_shape_dict(v, num_nodes, std, root=False, device=device)
            for k, v in parameter_shapes.items()
        })
    if isinstance(parameter_shapes, (int, tuple)):
        shape = (parameter_shapes,) if isinstance(parameter_shapes, int) else parameter_shapes
        value = std * torch.randn(num_nodes, *shape, device=device)
        return {"parameter": value} if root else nn.Parameter(value)
    raise Exception(f"Parameter shapes with type {type(parameter_shapes)} is not suppoerted")


def permute_grafx_tensor(G_t, node_id, node_attrs=("node_types", "rendering_orders"), id_attrs=("edge_indices",)):
    """Relabel nodes: node ``i`` becomes ``node_id[i]`` (utils.py:134-174)."""
    where_from = torch.empty_like(node_id)
    where_from[node_id] = torch.arange(len(node_id))
    fields = {}
    for k, v in G_t.__dict__.items():
        if v is None:
            fields[k] = None
        elif k in node_attrs:
            fields[k] = v[where_from]
        elif k in id_attrs:
            fields[k] = node_id[v]
        else:
            fields[k] = v
    return GRAFXTensor(**fields)
