"""`python bench.py --gpus N` must start N ranks by itself (the driver runs exactly that command shape).

CPU plumbing check: `--dry --backend gloo` swaps the HIP processors for pass-through modules on CPU tensors, so the
launcher, the rendezvous on 127.0.0.1, the batch sharding, the barrier + max-over-ranks timing and the JSON line are
exercised without a GPU.  (The numbers of a dry line mean nothing, and the line says so.)"""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*flags, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_launches_two_ranks():
    res = _run("--gpus", "2", "--dry", "--backend", "gloo", "--batch", "2", "--length", "4096", "--steps", "2",
               "--warmup", "1")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["backend"] == "gloo"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in out["per_rank_ms_per_step"])
    assert out["config"]["global_batch"] == 4 and out["config"]["batch_per_gpu"] == 2
    assert out["scaling"] == "weak" and out["steps"] == 2 and out["warmup"] == 1
    assert abs(out["ms_per_step"] - max(out["per_rank_ms_per_step"])) < 1e-9  # the job's time is the MAX over ranks
    assert "dry" in out
    # the training leg runs in the dry mode too: backward through the render + ONE flat all-reduce; every rank renders a
    # different shard (seeded by rank), so equal gradients afterwards mean the reduction really happened
    tr = out["training"]
    assert "error" not in tr, tr
    assert tr["grad_floats"] > 0 and tr["grad_sync"]["ranks"] == 2
    assert tr["grad_sync"]["grad_abs_sum"] > 0 and tr["grad_sync"]["max_abs_diff_across_ranks"] == 0.0
    # each rank reports the process group it joined (what a SCALE log will show for RCCL)
    assert res.stderr.count("torch.distributed backend=gloo world_size=2") == 2, res.stderr[-2000:]


def test_bench_gpus_8_dry_run_is_eight_ranks_eight_shards_one_gradient():
    """The node the driver's SCALE run wants (8 ranks) has never been available: everything but RCCL itself is exercised
    here -- eight processes, eight different shards of the batch, identical gradients after the flat all-reduce."""
    res = _run("--gpus", "8", "--dry", "--backend", "gloo", "--batch", "1", "--length", "2048", "--steps", "1",
               "--warmup", "0", timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["backend"] == "gloo"
    assert out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "batch-shard x8"
    assert len(out["per_rank_ms_per_step"]) == 8
    assert len(set(out["shard_fingerprints"])) == 8, out["shard_fingerprints"]
    tr = out["training"]
    assert "error" not in tr, tr
    assert tr["grad_sync"]["ranks"] == 8 and tr["grad_sync"]["grad_abs_sum"] > 0
    assert tr["grad_sync"]["max_abs_diff_across_ranks"] == 0.0
    assert res.stderr.count("torch.distributed backend=gloo world_size=8") == 8, res.stderr[-2000:]


def test_bench_single_rank_dry():
    res = _run("--dry", "--backend", "gloo", "--batch", "1", "--length", "2048", "--steps", "1", "--warmup", "0")
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["world_size"] == 1


def test_bench_launcher_propagates_a_failing_rank():
    # rank 1 of 2 cannot join (nccl without --dry on a box with no GPU / a bad flag combination): non-zero exit, no JSON
    res = _run("--gpus", "2", "--backend", "gloo", "--batch", "1", "--length", "2048", "--steps", "1", "--warmup", "0")
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 command: `python -m torch.distributed.run ... bench.py --gpus N` (ranks come from the env)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dry", "--backend", "gloo", "--batch", "1", "--length", "2048", "--steps", "1",
                          "--warmup", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2
