"""Render metadata (mirrors grafx.render.prepare — reference src/grafx/render/prepare.py:10-244).

``prepare_render`` turns a render-ordered ``GRAFXTensor`` into one
``_SingleRenderData`` per scheduled type: where to read the inputs from the
signal buffer (slice or index), how to aggregate multiple incoming edges
(none / sum / scatter), which parameter rows to read and where to write.
Index arithmetic is int64 and must match the reference bit for bit
(tests/test_routing_golden.py).
"""
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch

from ._index_ops import sort_edges_by_dest

TENSOR_IDX_TYPE = Union[Tuple[int], torch.LongTensor]


@dataclass
class _TensorAccessData:
    method: str
    idx: TENSOR_IDX_TYPE

    def __str__(self):
        return f"{self.method} with {self.idx}"


@dataclass
class _AggregationData:
    method: str
    idx: Optional[TENSOR_IDX_TYPE] = None

    def __str__(self):
        return self.method if self.method != "scatter" else f"scatter with {self.idx}"


@dataclass
class _SingleRenderData:
    node_type: str
    source_reads: List[_TensorAccessData]
    aggregations: List[_AggregationData]
    parameter_read: _TensorAccessData
    dest_write: _TensorAccessData

    def __str__(self):
        def block(title, items):
            if len(items) == 1:
                return [f"- {title}: {items[0]}"]
            return [f"- {title}s:"] + [f"  * {it}" for it in items]

        lines = [f"- Node type: {self.node_type}"]
        lines += block("Source read", self.source_reads)
        lines += block("Aggregation", self.aggregations)
        lines += [f"- Parameter read: {self.parameter_read}", f"- Dest write: {self.dest_write}"]
        return "\n".join(lines)


@dataclass
class RenderData:
    method: str
    num_nodes: int
    max_order: int
    siso_only: bool
    iter_list: List[_SingleRenderData]

    def __str__(self):
        head = f"Rendering of {self.num_nodes} nodes with siso_only: {self.siso_only}."
        return "\n\n".join([head] + [f"Render #{i}\n{it}" for i, it in enumerate(self.iter_list)])

    def to(self, device):
        """Move every index tensor to ``device`` once, so the render loop never syncs on host indices."""
        for it in self.iter_list:
            for acc in list(it.source_reads) + list(it.aggregations) + [it.parameter_read, it.dest_write]:
                if isinstance(acc.idx, torch.Tensor):
                    acc.idx = acc.idx.to(device)
        return self


def prepare_render(G_t):
    cfg = G_t.config
    siso = cfg.siso_only
    type_sequence = G_t.type_sequence
    row_in_type = create_per_type_indices(G_t.node_types)

    if siso:
        edges = sort_edges_by_dest(G_t.edge_indices)
        edge_ports = None
    else:
        edges, edge_ports = sort_edges_by_dest(G_t.edge_indices, G_t.edge_types)
        edge_ports = edge_ports.tolist()
        outlets_of = torch.tensor([cfg.num_outlets[t] for t in cfg.node_types])[G_t.node_types].tolist()
        slot_of = torch.cumsum(torch.tensor([0] + outlets_of[:-1]), 0).tolist()
    edges = edges.T

    max_order = torch.max(G_t.rendering_orders)
    iter_list = []
    for order in range(max_order + 1):
        member = G_t.rendering_orders == order
        members = torch.where(member)[0]
        member_list = members.tolist()
        node_type = type_sequence[order]
        incoming = get_incoming_edges(edges, members).tolist()

        if siso:
            sources = [s for s, _ in incoming]
            slots = [member_list.index(d) for _, d in incoming]
            source_reads = [check_and_convert_arange(sources)]
            aggregations = [check_aggregate_method(slots, member_list)]
        else:
            n_in = cfg.num_inlets[node_type]
            sources = [[] for _ in range(n_in)]
            slots = [[] for _ in range(n_in)]
            for s, d in incoming:
                # the reference indexes edge_types with the *order* index here (prepare.py:155); kept.
                outlet, inlet = edge_ports[order]
                slots[inlet].append(member_list.index(d))
                sources[inlet].append(slot_of[s] + outlet)
            source_reads = [check_and_convert_arange(s) for s in sources]
            aggregations = [check_aggregate_method(s, member_list) for s in slots]

        parameter_read = check_and_convert_arange(row_in_type[member])

        if siso:
            dest = list(member_list)
        else:
            width = cfg.num_outlets[node_type]
            dest = torch.tensor([slot_of[n] + k for n in member_list for k in range(width)])
        dest_write = check_and_convert_arange(dest)

        iter_list.append(_SingleRenderData(node_type=node_type, source_reads=source_reads, aggregations=aggregations,
                                           parameter_read=parameter_read, dest_write=dest_write))

    return RenderData(method=G_t.rendering_order_method, num_nodes=G_t.num_nodes, max_order=max_order,
                      siso_only=siso, iter_list=iter_list)


def check_aggregate_method(scatter_idx, node_list):
    """none / sum / scatter classification (reference prepare.py:198-215)."""
    if len(scatter_idx) == 0:
        return _AggregationData(method="none")
    idx = scatter_idx if isinstance(scatter_idx, torch.Tensor) else torch.tensor(scatter_idx)
    if len(idx) == 1 and idx[0] == 0:
        return _AggregationData(method="none")
    if (idx == 0).all():
        return _AggregationData(method="sum")
    identity = len(idx) == len(node_list) and idx[0] == 0 and not (idx.diff() != 1).any()
    if identity:
        return _AggregationData(method="none")
    return _AggregationData(method="scatter", idx=idx)


def check_and_convert_arange(idx):
    """A run of consecutive ids becomes a slice (a, b); anything else an index tensor (prepare.py:218-228)."""
    if len(idx) == 0:
        return _TensorAccessData(method="none", idx=idx)
    t = idx if isinstance(idx, torch.Tensor) else torch.tensor(idx)
    if (t.diff() == 1).all():
        return _TensorAccessData(method="slice", idx=(t[0].item(), t[-1].item() + 1))
    return _TensorAccessData(method="index", idx=t)


def get_incoming_edges(edges, node_idxs):
    hits = (edges[:, 1][:, None] == node_idxs[None, :]).any(-1)
    return edges[hits]


def create_per_type_indices(node_types):
    """Running index of each node within its own type (prepare.py:237-244)."""
    out = torch.zeros_like(node_types)
    for t in set(node_types.tolist()):
        mask = node_types == t
        out[mask] = torch.arange(int(mask.sum()))
    return out
