#!/usr/bin/env python3
"""Headline benchmark: forward render of the 32-channel mixing-console graph
(111 nodes: 36 EQ, 36 compressor, 1 reverb, 5 mix, 32 in, 1 out — BASELINE.json configs[3])
at batch 256 per GPU, L = 131072 samples, stereo, fp32.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one render_grafx pass over one resident batch.  The batch axis shards over GPUs
(weak scaling: 256 graphs per GPU, no data-path collective for the forward render).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 5.2-5.5 TB/s is what a streaming copy reaches (tools/ubench/stream.hip)


def console_graph(n_ch=32, n_bus=4):
    from grafx_amd.data import GRAFX, NodeConfigs

    G = GRAFX(config=NodeConfigs(["eq", "compressor", "reverb"]))
    out_id = G.add("out")
    buses = [G.add("mix") for _ in range(n_bus)]
    send = G.add("mix")
    for ch in range(n_ch):
        _, last = G.add_serial_chain(["in", "eq", "compressor"])
        G.connect(last, buses[ch // (n_ch // n_bus)])
        G.connect(last, send)
    for b in buses:
        e, c = G.add("eq"), G.add("compressor")
        G.connect(b, e)
        G.connect(e, c)
        G.connect(c, out_id)
    r = G.add("reverb")
    G.connect(send, r)
    G.connect(r, out_id)
    return G


# filter lengths: odd, so that L + N - 1 is even and the reference's convolve() IS the linear
# convolution (see DESIGN.md "length-parity quirk"); all are legal reference constructor arguments.
LENS = dict(fsm_fir_len=4001, iir_len=16383, ir_len=60001)


def hip_processors(default_args=False):
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb

    if default_args:  # upstream's constructor defaults: flashfftconv=True, 4000 / 16384 / 60000 taps
        return {"eq": ParametricEqualizer(num_filters=6), "compressor": Compressor(energy_smoother="iir"),
                "reverb": STFTMaskedNoiseReverb()}
    return {
        "eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=LENS["fsm_fir_len"]),
        "compressor": Compressor(energy_smoother="iir", iir_len=LENS["iir_len"], flashfftconv=False),
        "reverb": STFTMaskedNoiseReverb(ir_len=LENS["ir_len"], flashfftconv=False),
    }


def oracle_processors():
    import oracle

    return {
        "eq": oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=LENS["fsm_fir_len"]),
        "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=LENS["iir_len"]),
        "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=LENS["ir_len"]),
    }


def cpu_baseline(G, render_data, params_cpu, L, budget_s=25.0):
    """The CPU oracle (a port of the reference's algorithm) on the host cores, bounded sample.

    torch's CPU FFT does not scale to every core of a large host (oversubscription makes it slower),
    so a few thread counts are tried and the best one is reported with the count actually used."""
    from grafx_amd.render import render_grafx

    procs = oracle_processors()
    B = 8
    x = torch.randn(B, 32, 2, L)
    ncpu = os.cpu_count() or 1
    best, best_threads, runs = None, None, 0
    t_all = time.perf_counter()
    with torch.no_grad():
        for threads in sorted({min(ncpu, t) for t in (16, 32, 64)}):
            torch.set_num_threads(threads)
            for i in range(3):
                t0 = time.perf_counter()
                render_grafx(procs, x, params_cpu, render_data, parameters_grad=False)
                dt = time.perf_counter() - t0
                runs += 1
                if i > 0 and (best is None or dt < best):
                    best, best_threads = dt, threads
                if time.perf_counter() - t_all > budget_s:
                    break
            if time.perf_counter() - t_all > budget_s:
                break
    if best is None:
        best, best_threads = dt, threads
    return {
        "value": B * L / best,
        "unit": "audio samples/s",
        "cores": best_threads,
        "host_cpus": ncpu,
        "kind": "port",
        "sample": f"same 111-node console graph and filter lengths, batch {B} (of 256), L={L}; best of {runs} timed "
                  f"renders over thread counts 16/32/64 (torch CPU oracle, fp32)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="graphs per GPU")
    ap.add_argument("--length", type=int, default=131072)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reference-default-lengths", action="store_true",
                    help="use the reference's default (even) filter lengths 4000/16384/60000: every convolve() then takes "
                         "the odd-P aliasing compatibility path (DESIGN.md section 2); not the headline configuration")
    ap.add_argument("--reference-default-args", action="store_true",
                    help="build the processors with upstream's constructor defaults (flashfftconv=True, 4000/16384/60000 taps): "
                         "the FlashFFTConv flavour, i.e. plain causal convolutions; not the headline configuration")
    ap.add_argument("--capture", action="store_true",
                    help="replay the render as one captured HIP graph (grafx_amd.render.CapturedRender): the serving "
                         "path for small batches, where the eager loop is host-bound")
    ap.add_argument("--train", action="store_true",
                    help="also time forward+backward+gradient all-reduce (BASELINE configs[4]) at --train-batch per GPU")
    ap.add_argument("--train-batch", type=int, default=256, help="graphs per GPU in the training step (configs[4]: 256)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one process per GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        # RCCL builds its communicator (device buffers, proxy threads) lazily at the first collectives; do that now,
        # not inside the timed region (its hipMallocs are device-synchronising: seen as 250 ms "steps" with --warmup 1)
        warm = torch.zeros(1, device=torch.device("cuda", local))
        dist.all_reduce(warm)
        dist.barrier()
        torch.cuda.synchronize()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    if args.reference_default_lengths or args.reference_default_args:
        LENS.update(fsm_fir_len=4000, iir_len=16384, ir_len=60000)
    from grafx_amd import ops
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    B, L = args.batch, args.length
    G = console_graph()
    render_data = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    procs = {k: v.to(dev) for k, v in hip_processors(args.reference_default_args).items()}
    torch.manual_seed(1234)  # identical parameters on every rank (a shared mixing console)
    params_cpu = {t: {k: v.detach() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
    params = {t: {k: v.to(dev) for k, v in d.items()} for t, d in params_cpu.items()}
    torch.manual_seed(1000 + rank)  # each rank renders its own shard of the batch
    x = torch.randn(B, 32, 2, L, device=dev)
    rd_dev = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)

    captured = []

    def step():
        if captured:
            return captured[0]()[0]
        with torch.no_grad():
            return render_grafx(procs, x, params, rd_dev, parameters_grad=False)[0]

    # prime the caching allocator with the two signal buffers the loop ping-pongs between (the render
    # returns views of its (B, 111, 2, L) buffer, so one stays alive while the next step allocates):
    # a first hipMalloc of ~30 GB on a freshly booted box can take most of a second.
    pool = [torch.empty(B, render_data.num_nodes, 2, L, device=dev) for _ in range(2)]
    del pool
    with torch.no_grad():  # one single-graph render: loads every kernel's code object and builds the per-device tables
        render_grafx(procs, x[:1], params, rd_dev, parameters_grad=False)
    torch.cuda.synchronize()
    if args.capture:
        from grafx_amd.render import CapturedRender

        captured.append(CapturedRender(procs, x, params, rd_dev))
    y = None
    if dist is not None:
        # line the ranks up BEFORE the warm-up: their start-up skew (seconds) then is not spent idling at the barrier
        # that opens the timed region, right in front of the first timed step (a GPU that has idled for milliseconds
        # runs its next large launches 20-30 % slow)
        torch.cuda.synchronize()
        dist.barrier()
    for _ in range(args.warmup):
        y = step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    # the dominant kernel is timed live, inside the timed region, with a pair of stream events around every launch
    # (recorded on the stream the kernel runs on; no synchronisation, ~120 event records per 13 ms step)
    ops.PROFILE = None if args.capture else {}
    t0 = time.perf_counter()
    debug = os.environ.get("GRAFX_BENCH_DEBUG")
    for _ in range(args.steps):
        y = step()
        if debug:  # per-step wall times (adds a sync per step: diagnostics only)
            torch.cuda.synchronize()
            print(f"[bench] step done at +{(time.perf_counter() - t0) * 1e3:.1f} ms, reserved "
                  f"{torch.cuda.memory_reserved() / 2**30:.1f} GiB", file=sys.stderr)
    torch.cuda.synchronize()          # this rank's K steps are done: stop its clock ...
    elapsed = time.perf_counter() - t0
    fence()                           # ... then the closing barrier; the job's time is the MAX over ranks (below)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(y).all(), "render produced non-finite samples"

    roof = None
    prof, ops.PROFILE = ops.PROFILE, None
    if prof:
        stats = {}
        for name, recs in prof.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            stats[name] = (sum(ms), sum(ms) / len(ms), sum(r[2] for r in recs) / len(recs), len(ms))
        name = max(stats, key=lambda k: stats[k][0])
        total_ms, avg_ms, avg_bytes, n = stats[name]
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r1", "pmc_hbm_traffic.json")
        if os.path.exists(pmc):  # PMC bytes come from a separate rocprofv3 --pmc pass of this same command
            with open(pmc) as f:
                rec = json.load(f)
            cfg = rec.get("config", {})
            if cfg.get("batch") == B and cfg.get("audio_len") == L and all(cfg.get(k) == v for k, v in LENS.items()):
                k = rec["kernels"].get("gfx::" + name)
                if k:
                    traffic, traffic_src = k["hbm_bytes_per_launch"], "profiles/r1/pmc_hbm_traffic.json"
        roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": avg_bytes, "launches_per_step": n // args.steps,
                "share_of_step": total_ms / args.steps / (elapsed / args.steps * 1e3)}

    train = None
    if args.train:
        # BASELINE configs[4]: shared parameters, batch sharded over the GPUs, one flat gradient all-reduce
        import torch.nn as nn

        from grafx_amd.parallel import all_reduce_gradients

        del y
        torch.cuda.empty_cache()
        Bt = args.train_batch
        tparams = nn.ParameterDict({t: nn.ParameterDict({k: nn.Parameter(v.clone()) for k, v in d.items()})
                                    for t, d in params.items()})
        plist = list(tparams.parameters())
        xt = x[:Bt]

        def train_step():
            for p in plist:
                p.grad = None
            out = render_grafx(procs, xt, tparams, rd_dev)[0]
            out.square().mean().backward()
            all_reduce_gradients(plist)

        train_step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            train_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        fence()
        if dist is not None:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        train = {"what": "forward + backward + flat all-reduce of shared-parameter gradients", "batch_per_gpu": Bt,
                 "ms_per_step": dt / args.steps * 1e3, "value": world * Bt * L * args.steps / dt,
                 "unit": "audio samples/s", "grad_floats": sum(p.numel() for p in plist),
                 "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2**30}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            "metric": "audio samples/sec (rendered output frames of the 32-channel console graph, all GPUs)",
            "value": world * B * L * args.steps / elapsed,
            "unit": "audio samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (randn signals, randn*0.1 parameters, random-phase noise IRs)",
            "config": {"workload": "BASELINE configs[3]: 32-channel mixing console, 111 nodes / 142 edges, "
                                   "ParametricEqualizer(6) + Compressor(iir) + STFTMaskedNoiseReverb + bus sums",
                       "batch_per_gpu": B, "global_batch": B * world, "audio_len": L, "channels": 2,
                       "fsm_fir_len": LENS["fsm_fir_len"], "iir_len": LENS["iir_len"], "ir_len": LENS["ir_len"],
                       "mode": ("forward render, reference-default even lengths (odd L+N-1: aliasing compatibility path)"
                                if args.reference_default_lengths else
                                "forward render, upstream default constructor arguments (flashfftconv=True: plain causal convolutions)"
                                if args.reference_default_args else "forward render, reference-exact lengths (even L+N-1)"), "parallelism": f"batch-shard x{world}",
                       "launch": "one captured HIP graph per step" if args.capture else "eager render loop"},
            "per_gpu_value": B * L * args.steps / elapsed,
            "roofline": roof,
        }
        # whole-graph view of the same roofline: 285 row transfers per graph
        graph_bytes = 285 * B * 2 * L * 4
        out["graph_roofline"] = {"algorithmic_bytes_per_step": graph_bytes,
                                 "achieved_GBps": graph_bytes / (ms_per_step * 1e-3) / 1e9,
                                 "frac_of_hbm_peak": graph_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if train is not None:
            out["training"] = train
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(G, render_data, params_cpu, L)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
