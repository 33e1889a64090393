"""Signal-buffer plumbing for render_grafx (mirrors grafx.render.core —
reference src/grafx/render/core.py:6-140)."""
import torch
import torch.nn as nn

from ._index_ops import segment_reduce


def create_signal_buffer(method, num_buffers, input_signals):
    """Buffer holding every node's output; sources occupy the first slots (core.py:6-33).
    Always default-dtype (float32), as upstream."""
    if method == "one-by-one":
        return [x[None, :, :] for x in list(input_signals)] + [None] * (num_buffers - len(input_signals))
    device = input_signals.device
    if input_signals.ndim == 3:
        n_src, channels, length = input_signals.shape
        buf = torch.empty(num_buffers, channels, length, device=device)
        buf[:n_src] = input_signals
    else:
        batch, n_src, channels, length = input_signals.shape
        buf = torch.empty(batch, num_buffers, channels, length, device=device)
        buf[:, :n_src] = input_signals
    return buf


def read_single_tensor(x, access, dim=0, return_copy=False, postprocess=None):
    if access.method == "slice":
        start, stop = access.idx
        x = x.narrow(dim, start, stop - start)
        if return_copy:
            x = x.clone()
    elif access.method == "index":
        x = x.index_select(dim, access.idx)
    else:
        raise Exception(f"The provided read method is not available: {access.method}.")
    return x if postprocess is None else postprocess(x)


def read_tensor_or_tensor_dict(x, access, dim=0, return_copy=False, postprocess=None):
    if isinstance(x, torch.Tensor):
        return read_single_tensor(x, access, dim=dim, return_copy=return_copy, postprocess=postprocess)
    if isinstance(x, (dict, nn.ParameterDict, nn.ModuleDict)):
        return {k: read_tensor_or_tensor_dict(v, access, dim=dim, return_copy=return_copy, postprocess=postprocess)
                for k, v in x.items()}
    if isinstance(x, list):
        return x[access.idx[0]]
    return None


def read_tensor_dict(x, access, dim=0, return_copy=False):
    """Flat dict of tensors -> the same read applied to every value (reference render/core.py:73-77, which calls a
    ``read_tensor`` that does not exist there; the single-tensor read is what it means)."""
    return {k: read_single_tensor(v, access, dim=dim, return_copy=return_copy) for k, v in x.items()}


def inplace_write_tensor(method, x, y, access, dim=0):
    if method == "one-by-one":
        x[access.idx[0]] = y
        return
    if access.method == "slice":
        sel = slice(access.idx[0], access.idx[1])
    elif access.method == "index":
        sel = access.idx
    else:
        raise Exception(f"The provided inplace write method is not available: {access.method}.")
    if dim == 0:
        x[sel] = y
    elif dim == 1:
        x[:, sel] = y


def aggregate_tensor(x, aggregation, dim=0):
    if aggregation.method == "sum":
        return torch.sum(x, dim, keepdim=True)
    if aggregation.method == "scatter":
        return segment_reduce(x, aggregation.idx, dim=dim, reduce="sum")
    if aggregation.method == "none":
        return x
    raise Exception(f"The provided aggregation method is not available: {aggregation.method}.")


def expand_single_tensor(x, expand=2, dim=0):
    x = x.unsqueeze(dim)
    sizes = list(x.shape)
    sizes[dim] *= expand
    return x.expand(*sizes).contiguous()


def expand_tensor_or_tensor_dict(x, expand=2, dim=0):
    if isinstance(x, torch.Tensor):
        return expand_single_tensor(x, expand=expand, dim=dim)
    if isinstance(x, (dict, nn.ParameterDict, nn.ModuleDict)):
        return {k: expand_tensor_or_tensor_dict(v, expand=expand, dim=dim) for k, v in x.items()}
    return {}


def flatten_batch_and_node(x):
    return x.reshape(-1, *x.shape[2:])
