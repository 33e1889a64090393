"""Index helpers the render path needs (own implementations; the reference
borrows them from torch_geometric, which is not a dependency here —
call sites: reference src/grafx/render/core.py:106, prepare.py:115-119,
order/tensor.py:94,159,198)."""
import torch


def segment_reduce(src, index, dim=-1, dim_size=None, reduce="sum"):
    """out[..., j, ...] = reduce over {i : index[i] == j} of src[..., i, ...] along ``dim``.

    Empty segments give 0 for sum/min and 1 for mul (the only reductions the
    render path uses).
    """
    dim = dim % src.ndim
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    view = [1] * src.ndim
    view[dim] = -1
    idx = index.view(view).expand_as(src)
    shape = list(src.shape)
    shape[dim] = dim_size
    if reduce == "sum":
        return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_add_(dim, idx, src)
    if reduce == "mul":
        return torch.ones(shape, dtype=src.dtype, device=src.device).scatter_reduce_(dim, idx, src, "prod", include_self=True)
    if reduce == "min":
        return torch.zeros(shape, dtype=src.dtype, device=src.device).scatter_reduce_(dim, idx, src, "amin", include_self=False)
    raise ValueError(f"unsupported reduce: {reduce}")


def sort_edges_by_dest(edge_indices, edge_attr=None):
    """Stable sort of a [2,E] edge list by (dest, source)."""
    if edge_indices.numel() == 0:
        return edge_indices if edge_attr is None else (edge_indices, edge_attr)
    n = int(edge_indices.max()) + 1
    perm = torch.argsort(edge_indices[1] * n + edge_indices[0], stable=True)
    if edge_attr is None:
        return edge_indices[:, perm]
    return edge_indices[:, perm], edge_attr[perm]
