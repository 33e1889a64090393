"""Index-only semantics of the two torch_geometric.utils functions the reference calls."""
import torch


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    dim = dim % src.ndim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = [1] * src.ndim
    shape[dim] = -1
    idx = index.view(shape).expand_as(src)
    out_shape = list(src.shape)
    out_shape[dim] = dim_size
    if reduce in ("sum", "add"):
        out = torch.zeros(out_shape, dtype=src.dtype, device=src.device)
        return out.scatter_add_(dim, idx, src)
    red = {"mul": "prod", "min": "amin", "max": "amax", "mean": "mean"}[reduce]
    init = {"prod": 1, "amin": 0, "amax": 0, "mean": 0}[red]
    out = torch.full(out_shape, init, dtype=src.dtype, device=src.device)
    return out.scatter_reduce_(dim, idx, src, red, include_self=(red == "prod"))


def sort_edge_index(edge_index, edge_attr=None, sort_by_row=True):
    num_nodes = int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
    major = edge_index[1 - int(sort_by_row)]
    minor = edge_index[int(sort_by_row)]
    perm = torch.argsort(major * num_nodes + minor, stable=True)
    edge_index = edge_index[:, perm]
    if edge_attr is None:
        return edge_index
    return edge_index, edge_attr[perm]
