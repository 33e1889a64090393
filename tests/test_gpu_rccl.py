"""The RCCL call site of the training path on a real GPU: a 1-rank `nccl` process group (this box has one GPU) forced
through the flat gradient all-reduce of grafx_amd.parallel -- the same flatten / all_reduce / scatter-back sequence the
8-GPU job (BASELINE configs[4]) runs after every backward pass.  In-process (a GPU process must not exec children on
this pool); the group is destroyed again before the test returns."""
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_flat_gradient_all_reduce_runs_on_rccl_with_one_rank():
    import torch.distributed as dist

    from grafx_amd.data import convert_to_tensor
    from grafx_amd.parallel import all_reduce_gradients, gather_outputs, shard_batch
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters
    from test_routing_golden import build_console

    assert not dist.is_initialized()
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    real = dist.all_reduce
    try:
        assert dist.get_backend() == "nccl"
        torch.manual_seed(0)
        G = build_console(4, 2)
        procs = {"eq": ParametricEqualizer(num_filters=4, flashfftconv=False, fsm_fir_len=257).cuda(),
                 "compressor": Compressor(energy_smoother="iir", iir_len=255, flashfftconv=False).cuda(),
                 "reverb": STFTMaskedNoiseReverb(ir_len=1501, flashfftconv=False).cuda()}
        rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
        params = create_empty_parameters(procs, G, std=0.2).cuda()
        x = shard_batch(torch.randn(3, 4, 2, 4096, device=dev))
        assert x.shape[0] == 3
        y = render_grafx(procs, x, params, rd)[0]
        y.square().mean().backward()
        plist = list(params.parameters())
        before = [p.grad.clone() for p in plist]
        calls = []
        dist.all_reduce = lambda t, *a, **k: (calls.append((t.numel(), t.device.type)), real(t, *a, **k))[1]
        all_reduce_gradients(plist)                      # world size 1: returns early, no collective
        assert not calls
        all_reduce_gradients(plist, force=True)          # ... forced: ONE flat all-reduce, on RCCL
        torch.cuda.synchronize()
        assert calls == [(sum(g.numel() for g in before), "cuda")], calls
        for g, p in zip(before, plist):
            assert torch.equal(g, p.grad)                # the sum over one rank, averaged by one
        assert gather_outputs(y) is y
        dist.barrier()
    finally:
        dist.all_reduce = real
        dist.destroy_process_group()
        # RCCL writes its version banner through C stdio; flushed only at process exit it would land BEHIND pytest's
        # summary line -- flush it now, while this test's output is still being captured
        import ctypes

        ctypes.CDLL(None).fflush(None)
    assert not dist.is_initialized()
