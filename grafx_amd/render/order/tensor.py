"""Type scheduling on GRAFXTensor (mirrors grafx.render.order.tensor —
reference src/grafx/render/order/tensor.py:12-247).

Given node types and edges, find a sequence of node *types* such that
rendering "all currently computable nodes of type t" once per step covers the
graph in few steps.  Results (type_sequence, per-node order) must equal the
reference's bit for bit, so candidate scoring and tie-breaking use the same
tensor ops on the same flattened layouts (argmax / argsort(descending)).
"""
from typing import List

import torch

from ...utils import permute_grafx_tensor
from .._index_ops import segment_reduce

MAX_ITER = 100
_IN, _OUT = 0, 1  # indices of the utility types "in" / "out" in every NodeConfigs


def return_render_ordered_tensor(G_t, method, **kwargs):
    type_sequence, render_order = compute_render_order_tensor(G_t, method, **kwargs)
    G_t.type_sequence = [G_t.config.node_types[t] for t in type_sequence]
    G_t.rendering_orders = render_order
    G_t.rendering_order_method = method
    return permute_grafx_tensor(G_t, node_id_from_render_order(render_order))


@torch.no_grad()
def compute_render_order_tensor(G_t, method: str = "beam", **kwargs):
    if method == "greedy":
        return greedy_search(G_t, **kwargs)
    if method == "beam":
        return beam_search(G_t, **kwargs)
    if method == "fixed":
        return fixed_order_search(G_t, **kwargs)
    if method == "one-by-one":
        return one_by_one_search(G_t, **kwargs)
    raise Exception(f"Invalid rendering method: {method}.")


def _ready_nodes(done, src, dst, num_nodes):
    """Nodes not yet done whose every predecessor is done (nodes without predecessors count as ready)."""
    all_inputs_done = segment_reduce(done[..., src].long(), dst, dim=-1, dim_size=num_nodes, reduce="mul").bool()
    return ~done & all_inputs_done


def _processor_types(T):
    kinds = sorted(set(T.tolist()))
    assert _IN in kinds and _OUT in kinds
    return torch.tensor([k for k in kinds if k not in (_IN, _OUT)], dtype=torch.long, device=T.device)


def greedy_search(G_t):
    return beam_search(G_t, width=1, depth=1)


def beam_search(G_t, depth: int = 1, width: int = 64):
    """Beam search over type sequences (reference tensor.py:127-230; defaults depth=1, width=64)."""
    T, E, V = G_t.node_types, G_t.edge_indices, G_t.num_nodes
    src, dst = E[0], E[1]
    dev = T.device
    kinds = _processor_types(T)
    of_kind = T[None, :] == kinds[:, None]  # (K, V)

    if width > 1 and depth > 1:
        counter = torch.arange(width * len(kinds) ** depth, device=dev)

    order = -torch.ones(1, V, dtype=torch.long, device=dev)
    order[:, T == _IN] = 0
    sequences = torch.zeros(1, 1, dtype=torch.long, device=dev)
    done = ((T == _IN) + (T == _OUT))[None, :]  # (beams, V)

    for step in range(1, MAX_ITER + 1):
        # expand every beam by every type, optionally looking `depth` steps ahead
        look = done
        for d in range(depth):
            ready = _ready_nodes(look, src, dst, V)
            gained = of_kind * ready.unsqueeze(-2)  # (..., K, V)
            look = look.unsqueeze(-2) + gained
            score = torch.count_nonzero(look, -1)
            if d == 0:
                after_one, gained_one = look, gained
            if (score == V).any():
                break

        n_beams, n_kinds = score.shape[0], score.shape[1]
        score = score.view(n_beams, n_kinds, -1)
        fan = score.shape[-1]
        flat = score.reshape(-1)

        if width == 1:
            pick = torch.argmax(flat, keepdim=True)
            if depth > 1:
                pick = pick // fan
        elif depth == 1:
            pick = torch.argsort(flat, descending=True)[:width]
        else:
            ranked = torch.argsort(flat, descending=True) // fan
            uniq, inverse = torch.unique(ranked, return_inverse=True)
            first_seen = segment_reduce(counter[: len(ranked)], inverse, dim=-1, reduce="min")
            pick = uniq[torch.argsort(first_seen)][:width]

        beam, kind = pick // n_kinds, pick % n_kinds
        done = after_one[beam, kind]
        sequences = torch.cat([sequences[beam], kinds[kind][:, None]], -1)
        order = order[beam]
        order[gained_one[beam, kind]] = step

        finished = done.all(-1)
        if finished.any():
            break
        if step == MAX_ITER:
            raise AssertionError("render-order search did not terminate")

    best = torch.argmax(finished.long())
    sequence = torch.cat([sequences[best], torch.tensor([_OUT], device=dev)])
    order = order[best]
    order[T == _OUT] = step + 1
    return sequence, order


def fixed_order_search(G_t, fixed_order: List[int]):
    """Walk a user-given cyclic list of type ids, emitting a step whenever that type has ready nodes
    (reference tensor.py:65-120)."""
    T, E, V = G_t.node_types, G_t.edge_indices, G_t.num_nodes
    src, dst = E[0], E[1]
    dev = T.device
    _processor_types(T)  # same precondition check as upstream

    order = -torch.ones(V, dtype=torch.long, device=dev)
    order[T == _IN] = 0
    sequence = [_IN]
    done = (T == _IN) + (T == _OUT)

    cursor, step = 0, 1
    for _ in range(MAX_ITER):
        ready = _ready_nodes(done, src, dst, V)
        while True:
            cursor += 1
            kind = fixed_order[cursor]
            chosen = ready * (T == kind)
            if torch.any(chosen):
                done = done + chosen
                sequence.append(kind)
                order[chosen] = step
                step += 1
                break
        if done.all():
            break
        if cursor == MAX_ITER:
            raise AssertionError("fixed-order search did not terminate")

    sequence.append(_OUT)
    order[T == _OUT] = step
    return torch.tensor(sequence, device=dev), order


def one_by_one_search(G_t):
    """Greedy schedule unrolled to one node per step (reference tensor.py:39-62)."""
    g_sequence, g_order = greedy_search(G_t)
    dev = g_order.device
    order = -torch.ones(len(g_order), dtype=torch.long, device=dev)
    sequence, nxt, level = [], 0, 0
    while True:
        members = g_order == level
        if level == 0:
            order[members] = 0
            sequence.append(0)
            nxt = 1
        else:
            count = int(torch.count_nonzero(members))
            if count == 0:
                break
            order[members] = torch.arange(nxt, nxt + count, device=dev)
            sequence += [g_sequence[level].item()] * count
            nxt += count
        level += 1
    return sequence, order


def node_id_from_render_order(render_order):
    """New node ids that make same-order nodes contiguous, stable within an order (tensor.py:233-247)."""
    node_id = -torch.ones(len(render_order), dtype=torch.long, device=render_order.device)
    nxt, level = 0, 0
    while True:
        members = render_order == level
        count = int(torch.count_nonzero(members))
        if count == 0:
            break
        node_id[members] = torch.arange(nxt, nxt + count, device=render_order.device)
        nxt += count
        level += 1
    return node_id
