"""Multi-GPU use of the render path: one process per GPU, batch-of-graphs axis sharded.

The forward render needs no collective (graphs in a batch never exchange signals —
reference src/grafx/data/batch.py:34 builds a disconnected union).  For the training path the
only exchange is the sum of the (tiny, ~6 KB) shared-parameter gradients: one flat all-reduce
on RCCL over xGMI ("nccl" backend on ROCm), latency-bound.
"""
import torch
import torch.distributed as dist


def shard_batch(x, rank=None, world_size=None):
    """Contiguous shard of the leading (batch-of-graphs) axis for this rank."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    B = x.shape[0]
    per = (B + world_size - 1) // world_size
    return x[rank * per : min(B, (rank + 1) * per)]


def all_reduce_gradients(parameters, average=True, force=False):
    """Sum (or average) the gradients of shared parameters across ranks with ONE flat all-reduce.  A single rank has
    nothing to exchange and returns at once; ``force=True`` runs the collective anyway (a 1-rank RCCL all-reduce is an
    identity, but it is the same flatten / all_reduce / scatter-back call sequence: tests/test_gpu_rccl.py)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads or not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    offset = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[offset : offset + n].view_as(g))
        offset += n


def gather_outputs(y, dst=0):
    """Concatenate per-rank output shards on ``dst`` (None elsewhere).

    ``shard_batch`` uses a ceil split, so shards can be uneven (10 graphs over 4 ranks: 3/3/3/1) or empty; the
    shard sizes are exchanged first and every rank sends a buffer padded to the largest shard, because
    ``dist.gather`` needs equal shapes on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return y
    world = dist.get_world_size()
    n = torch.tensor([y.shape[0]], dtype=torch.int64, device=y.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    per = max(sizes)
    send = y.contiguous()
    if send.shape[0] != per:
        pad = torch.zeros((per - send.shape[0], *send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad])
    bucket = [torch.empty_like(send) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(send, bucket, dst=dst)
    return torch.cat([b[:k] for b, k in zip(bucket, sizes)]) if bucket is not None else None
