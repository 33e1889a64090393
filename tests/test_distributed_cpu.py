"""N>1 path on CPU: two gloo ranks shard the batch, render with the CPU oracle processors through
grafx_amd's render loop, all-reduce shared-parameter gradients with one flat collective."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.parallel import all_reduce_gradients, gather_outputs, shard_batch
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters
    from test_routing_golden import build_console

    torch.manual_seed(0)
    G = build_console(4, 2)
    procs = {"eq": oracle.OracleParametricEqualizer(num_filters=3, fsm_fir_len=65),
             "compressor": oracle.OracleCompressor(iir_len=63),
             "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=769)}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = create_empty_parameters(procs, G, std=0.2)  # same seed -> same parameters on every rank
    x_all = torch.randn(4, 4, 2, 512)
    x = shard_batch(x_all, rank, world)
    y, _, _ = render_grafx(procs, x, params, rd)
    y.square().sum().backward()
    all_reduce_gradients(list(params.parameters()), average=False)
    full = gather_outputs(y.detach(), dst=0)
    if rank == 0:
        torch.save({"y": full, "grads": [p.grad.clone() for p in params.parameters()]}, os.path.join(out_dir, "dist.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process(tmp_path):
    import oracle
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters
    from test_routing_golden import build_console

    port = 29600 + os.getpid() % 200
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    got = torch.load(os.path.join(tmp_path, "dist.pt"))

    torch.manual_seed(0)
    G = build_console(4, 2)
    procs = {"eq": oracle.OracleParametricEqualizer(num_filters=3, fsm_fir_len=65),
             "compressor": oracle.OracleCompressor(iir_len=63),
             "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=769)}
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    params = create_empty_parameters(procs, G, std=0.2)
    x_all = torch.randn(4, 4, 2, 512)
    y, _, _ = render_grafx(procs, x_all, params, rd)
    y.square().sum().backward()
    assert torch.allclose(got["y"], y.detach(), atol=1e-6)
    for a, p in zip(got["grads"], params.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-4, atol=1e-6)


def test_shard_batch_covers_everything():
    from grafx_amd.parallel import shard_batch

    x = torch.arange(10)[:, None]
    parts = [shard_batch(x, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts), x)


def _gather_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grafx_amd.parallel import gather_outputs, shard_batch

    x = torch.arange(5 * 3, dtype=torch.float32).view(5, 3)  # 5 graphs over 3 ranks: 2 / 2 / 1
    full = gather_outputs(shard_batch(x, rank, world) * 2.0, dst=0)
    y = torch.arange(2 * 3, dtype=torch.float32).view(2, 3)  # 2 graphs over 3 ranks: 1 / 1 / 0 (an empty shard)
    full2 = gather_outputs(shard_batch(y, rank, world) + 1.0, dst=0)
    if rank == 0:
        torch.save({"full": full, "full2": full2}, os.path.join(out_dir, "gather.pt"))
    else:
        assert full is None and full2 is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_outputs_uneven_and_empty_shards(tmp_path):
    port = 29850 + os.getpid() % 100
    mp.start_processes(_gather_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True, start_method="spawn")
    got = torch.load(os.path.join(tmp_path, "gather.pt"))
    assert torch.equal(got["full"], torch.arange(15, dtype=torch.float32).view(5, 3) * 2.0)
    assert torch.equal(got["full2"], torch.arange(6, dtype=torch.float32).view(2, 3) + 1.0)
