#!/usr/bin/env python3
"""Headline benchmark: forward render of the 32-channel mixing-console graph
(111 nodes: 36 EQ, 36 compressor, 1 reverb, 5 mix, 32 in, 1 out — BASELINE.json configs[3])
at batch 256 per GPU, L = 131072 samples, stereo, fp32.

    python bench.py [--gpus N --steps K --warmup W]                      # N > 1: starts N ranks itself
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A step = one render_grafx pass over one resident batch.  The batch axis shards over GPUs
(weak scaling: 256 graphs per GPU, no data-path collective for the forward render).
Prints ONE JSON line on rank 0.

`--gpus N` without a launcher (no RANK in the environment) makes this process a pure launcher: it starts
N children — one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1 — BEFORE
anything here touches the GPU, relays rank 0's JSON line and exits non-zero if any rank failed.

Other workloads (`--config`): cfg2 = BASELINE configs[1] (ParametricEqualizer, 1024 x 1 x 480000),
cfg3 = configs[2] (STFTMaskedNoiseReverb, 512 x 2 x 240000); each prints the same kind of line with its own
roofline.  `--dry --backend gloo` is the CPU plumbing check of the multi-rank path (pass-through processors,
no audio arithmetic; used by tests/test_bench_launch.py) — its numbers mean nothing and the line says so.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFS = 157.3   # MI355X vector FP32 (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFS = 78.6    # vector FP64: half the FP32 rate (AMD's MI355X figure; the guide's table does not list it)
HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); a one-float4-per-thread nt copy reaches 6.6 TB/s on this
                       # pool (profiles/r2/stream2_copy_ceiling.txt), the guide measured 6.29 TB/s


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="graphs (cfg4) or rows (cfg2/cfg3) per GPU")
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--config", choices=["cfg4", "cfg2", "cfg3"], default="cfg4",
                    help="cfg4: the headline console graph (default); cfg2 / cfg3: BASELINE configs[1] / configs[2]")
    ap.add_argument("--console-variant", choices=["headline", "longpole", "clamp", "ballistics"], default="headline",
                    help="cfg4 only, for profiles of the secondary legs: longpole = every compressor's smoother logit at 6 "
                         "(pole 0.9975), clamp = logit 12 (pole 1 - 1e-5: live truncation term), ballistics = "
                         "Compressor(energy_smoother='ballistics'); the line's config says so")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only with --dry)")
    ap.add_argument("--dry", action="store_true",
                    help="CPU plumbing check: pass-through processors on CPU tensors (launch, sharding, barriers, JSON)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 6 s repetition of the headline step (`sustained`)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the cfg2 / cfg3 measurements (BASELINE configs[1] / configs[2]) that the default single-GPU "
                         "run appends to the line as `secondary`, with their compat (upstream default tap counts) twins")
    ap.add_argument("--reference-default-lengths", action="store_true",
                    help="use the reference's default (even) filter lengths 4000/16384/60000: every convolve() then takes "
                         "the odd-P aliasing compatibility path (DESIGN.md section 2); not the headline configuration")
    ap.add_argument("--reference-default-args", action="store_true",
                    help="build the processors exactly as written for upstream (constructor defaults: flashfftconv=True, "
                         "4000/16384/60000 taps).  As on the reference's CPU path the flag falls back to the native convolve(), "
                         "so this is --reference-default-lengths reached through the default arguments")
    ap.add_argument("--capture", action="store_true",
                    help="replay the render as one captured HIP graph (grafx_amd.render.CapturedRender): the serving "
                         "path for small batches, where the eager loop is host-bound")
    ap.add_argument("--train", action="store_true", help="(kept for compatibility: the training step is timed by default)")
    ap.add_argument("--train-batch", type=int, default=256, help="graphs per GPU in the training step (configs[4]: 256)")
    ap.add_argument("--train-steps", type=int, default=3)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher: one child per GPU.  Nothing in this process has touched the GPU
    (torch is not even imported yet), so this is a plain fork+exec of fresh interpreters, never a re-exec of a process
    that holds the device."""
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    children = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GRAFX_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    try:
        pending = dict(enumerate(children))
        while pending:
            for rank, p in list(pending.items()):
                code = p.poll()
                if code is None:
                    continue
                del pending[rank]
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[bench] rank {rank} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending.values():  # exactly the PIDs started above
                        q.terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in children:
            p.terminate()
        rc = 130
    return rc


# ------------------------------------------------------------------------------------------------ workloads
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


# upstream's constructor defaults: even, so that L + N - 1 is odd at every BASELINE audio length and convolve() aliases
# (SURVEY F3): the "compat" legs of the line
REFERENCE_DEFAULT_LENS = dict(fsm_fir_len=4000, iir_len=16384, ir_len=60000)


def hip_processors(default_args=False, lens=None, energy_smoother="iir"):
    from grafx_amd.processors import Compressor, ParametricEqualizer, STFTMaskedNoiseReverb

    if default_args:  # upstream's constructor defaults: flashfftconv=True (-> warning + native convolve), 4000 / 16384 / 60000 taps
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")   # "FlashFFTConv is not available. Using native convolution instead."
            return {"eq": ParametricEqualizer(num_filters=6), "compressor": Compressor(energy_smoother="iir"),
                    "reverb": STFTMaskedNoiseReverb()}
    lens = lens or LENS
    return {
        "eq": ParametricEqualizer(num_filters=6, flashfftconv=False, fsm_fir_len=lens["fsm_fir_len"]),
        "compressor": Compressor(energy_smoother=energy_smoother, iir_len=lens["iir_len"], flashfftconv=False),
        "reverb": STFTMaskedNoiseReverb(ir_len=lens["ir_len"], flashfftconv=False),
    }


def dry_processors():
    """Pass-through stand-ins with the processors' interface, for the CPU plumbing check only."""
    import torch.nn as nn

    class Pass(nn.Module):
        """y = x * (1 + mean(p)): a pass-through at p = 0 whose one parameter per node still receives a gradient, so
        that the dry run exercises the training leg (backward + flat all-reduce) as well."""

        def forward(self, input_signals, p=None, **params):
            if p is None or not p.requires_grad:
                return input_signals
            return input_signals * (1.0 + p.reshape(p.shape[0], -1).mean(-1).view(-1, 1, 1))

        def parameter_size(self):
            return {"p": 1}

    return {"eq": Pass(), "compressor": Pass(), "reverb": Pass()}


def oracle_processors():
    import oracle

    return {
        "eq": oracle.OracleParametricEqualizer(num_filters=6, fsm_fir_len=LENS["fsm_fir_len"]),
        "compressor": oracle.OracleCompressor(energy_smoother="iir", iir_len=LENS["iir_len"]),
        "reverb": oracle.OracleSTFTMaskedNoiseReverb(ir_len=LENS["ir_len"]),
    }


def _timed_cpu(fn, budget_s, warm=2, reps=5, min_total_s=0.0):
    """median of `reps` (more when they are short: at least `min_total_s` of timed work) after `warm` warm-ups
    (SURVEY section 8d), cut short when the budget runs out"""
    t_all = time.perf_counter()
    times = []
    i = 0
    while True:
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if i >= warm:
            times.append(dt)
        i += 1
        spent = time.perf_counter() - t_all
        if i >= warm + 1 and spent > budget_s:
            break
        if len(times) >= reps and sum(times) >= min_total_s:
            break
    return times


def _best_cpu(run, budget_s, min_total_s=0.0, probe=None, sweep=(64, 128)):
    """SURVEY 8d asks for all host cores; torch's CPU FFT gets much SLOWER when a large host is oversubscribed (256
    threads on the GPU box: ~25x), so the sample is timed on min(cores, 32) threads, then -- one warm-up and two timed runs
    each -- on the thread counts of `sweep` the host has, and ONE un-warmed run of `probe` (a smaller sample) on all cores
    is reported next to them.  The fastest thread count is the baseline.
    -> (host cpus, threads of the best point, its times, all-core probe seconds, {threads: median seconds})"""
    import torch

    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    torch.set_num_threads(threads)
    times = _timed_cpu(run, budget_s, min_total_s=min_total_s)
    points = {threads: statistics.median(times)}
    best = (threads, times)
    for t in sweep:
        if t > ncpu or t in points:
            continue
        torch.set_num_threads(t)
        ts = _timed_cpu(run, budget_s / 3, warm=1, reps=2)
        points[t] = statistics.median(ts)
        if points[t] < statistics.median(best[1]):
            best = (t, ts)
    probe_s = None
    if probe is not None and ncpu > max(points):
        torch.set_num_threads(ncpu)
        t0 = time.perf_counter()
        probe()
        probe_s = time.perf_counter() - t0
    torch.set_num_threads(best[0])
    return ncpu, best[0], best[1], probe_s, points


def cpu_baseline_console(G, render_data, params_cpu, L, budget_s=15.0):
    """The CPU oracle (a port of the reference's algorithm) on the host cores, bounded sample."""
    import torch

    from grafx_amd.render import render_grafx

    procs = oracle_processors()
    B = 4
    x = torch.randn(B, 32, 2, L)

    def run():
        with torch.no_grad():
            render_grafx(procs, x, params_cpu, render_data, parameters_grad=False)

    def probe():
        with torch.no_grad():
            render_grafx(procs, x[:1], params_cpu, render_data, parameters_grad=False)

    ncpu, threads, times, probe_s, points = _best_cpu(run, budget_s, probe=probe)
    med = statistics.median(times)
    return {"value": B * L / med, "unit": "audio samples/s", "cores": threads, "host_cpus": ncpu, "kind": "port",
            "seconds": [round(t, 3) for t in times],
            "thread_sweep": {str(t): round(B * L / s_, 1) for t, s_ in sorted(points.items())},
            "all_cores_probe": None if probe_s is None else {
                "threads": ncpu, "value": L / probe_s, "seconds": round(probe_s, 2),
                "sample": "one un-warmed render of batch 1 on all host threads (oversubscribed torch FFT: slower)"},
            "sample": f"same 111-node console graph and filter lengths, batch {B} (of 256), L={L}; median of "
                      f"{len(times)} renders after 2 warm-ups, {threads} torch threads (torch CPU oracle, fp32)"}


def cpu_baseline_proc(kind, L, budget_s=24.0):
    import torch

    import oracle

    torch.manual_seed(0)
    if kind == "cfg2":
        R = 8
        proc = oracle.OracleParametricEqualizer(num_filters=6, processor_channel="mono", fsm_fir_len=LENS["fsm_fir_len"])
        x = torch.randn(R, 1, L)
        p = {k: torch.randn(R, 1, 6) for k in ("w0", "q_inv", "log_gain")}
        C = 1
    else:
        R = 32
        proc = oracle.OracleSTFTMaskedNoiseReverb(ir_len=LENS["ir_len"])
        x = torch.randn(R, 2, L)
        p = {k: torch.randn(R, 2, 193) for k in ("init_log_magnitude", "delta_log_magnitude")}
        C = 2

    def run():
        with torch.no_grad():
            proc(x, **p)

    ncpu, threads, times, _, _ = _best_cpu(run, budget_s, min_total_s=10.0, sweep=())
    med = statistics.median(times)
    return {"value": R * C * L / med, "unit": "channel-samples/s", "cores": threads, "host_cpus": ncpu, "kind": "port",
            "seconds": [round(t, 3) for t in times], "timed_seconds_total": round(sum(times), 2),
            "sample": f"{R} rows (of the full batch) x {C} x {L}, same processor and filter length; median of {len(times)} "
                      f"calls after 2 warm-ups, {threads} torch threads (torch CPU oracle, fp32)"}


def profile_traffic(kernel, B, L, lens):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/rN/pmc_hbm_traffic.json: separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 corrections applied) -- a PROFILE figure,
    not one measured in this run; None when no committed profile matches the workload."""
    for rnd in ("r6", "r5", "r4", "r3", "r2", "r1"):
        pmc = os.path.join(ROOT, "profiles", rnd, "pmc_hbm_traffic.json")
        if not os.path.exists(pmc):
            continue
        with open(pmc) as f:
            rec = json.load(f)
        cfg = rec.get("config", {})
        if cfg.get("batch") == B and cfg.get("audio_len") == L and all(cfg.get(k) == v for k, v in lens.items()):
            for key in (kernel, "gfx::" + kernel, "gfx::" + kernel.split("<")[0]):
                k = rec["kernels"].get(key)
                if k:
                    return k["hbm_bytes_per_launch"], f"profiles/{rnd}/pmc_hbm_traffic.json"
    return None, None


def roofline_from_profile(prof, steps, elapsed, B, L, lens=None):
    """Dominant kernel (largest summed launch time inside the timed region, HIP events on its stream).  Records are keyed
    by the kernels' own names -- what `rocprofv3 --kernel-trace` prints for the same launches."""
    stats = {}
    for name, recs in prof.items():
        ms = [a.elapsed_time(b) for a, b, _ in recs]
        stats[name] = (sum(ms), sum(ms) / len(ms), sum(r[2] for r in recs) / len(recs), len(ms))
    name = max(stats, key=lambda k: stats[k][0])
    total_ms, avg_ms, avg_bytes, n = stats[name]
    achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
    traffic, traffic_src = profile_traffic(name, B, L, lens or LENS)
    return {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_profile": traffic,
            "traffic_source": None if traffic_src is None else
            f"{traffic_src} (separate rocprofv3 --pmc passes of this command, not measured in this run)",
            "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": avg_bytes, "launches_per_step": n // steps,
            "share_of_step": total_ms / steps / (elapsed / steps * 1e3),
            "per_kernel_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(stats.items())},
            "per_kernel_frac_of_hbm_peak": {k: round(v[2] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                            for k, v in sorted(stats.items())}}


def czt_valu_roofline(prof, steps, chains):
    """The chirp-z chains of a compat leg against the VECTOR-ALU roofline (they are FFT work: HBM bytes are the wrong
    yardstick for them).  ``chains``: (precise?, rows per step, P) of every odd_alias call of a step.  Arithmetic model per
    PAIR of rows: two circular convolutions of NFFT points = 4 FFTs at 5 N log2 N real flops + 3 pointwise complex products
    at 6 N (the chirp factors, the two spectra), NFFT from the library's own plan (gfx_odd_alias_pair_workspace_bytes)."""
    import math

    from grafx_amd._lib import lib

    out = {}
    for precise in (False, True):
        tag = "<double>" if precise else "<float>"
        ms = sum(sum(a.elapsed_time(b) for a, b, _ in recs) for name, recs in prof.items() if name.startswith("czt") and name.endswith(tag))
        flops = 0.0
        for pr, rows, P in chains:
            if pr != precise:
                continue
            per_pair = lib().gfx_odd_alias_pair_workspace_bytes(2, P) - 256
            nfft = per_pair // 8
            flops += (rows / 2) * (4 * 5 * nfft * math.log2(nfft) + 3 * 6 * nfft)
        if ms > 0 and flops > 0:
            peak = FP64_VALU_PEAK_TFS if precise else FP32_VALU_PEAK_TFS
            tfs = flops * steps / (ms * 1e-3) / 1e12
            # what the chain MOVES: five sweeps of the pair's workspace (in: write; two tile passes and the fused middle
            # pass: read + write each; out: read) = 8 x NFFT complex points per pair, plus z in and y out -- the yardstick
            # for how well the sweeps run, next to the algorithmic bytes (z + y) the call rooflines above are priced on
            sweep = 0.0
            for pr, rows, P in chains:
                if pr != precise:
                    continue
                ws_pair = (lib().gfx_odd_alias_pair_workspace_bytes(2, P) - 256) * (2 if precise else 1)
                sweep += (rows / 2) * (8 * ws_pair + 2 * 4 * (2 * P - 1))
            gbps = sweep * steps / (ms * 1e-3) / 1e9
            out["double" if precise else "float"] = {"bound": "valu", "achieved": tfs, "peak": peak, "unit": "TFLOP/s",
                                                     "frac": tfs / peak, "ms_per_step": ms / steps,
                                                     "flops_per_step": flops,
                                                     "model": "per pair of rows: 4 FFTs of NFFT points at 5 N log2 N + 3 pointwise products at 6 N",
                                                     "sweeps": {"bound": "hbm", "moved_bytes_per_step": sweep, "achieved": gbps,
                                                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
                                                                "model": "five sweeps of the pair's workspace (8 x NFFT points moved per pair) + z in + y out"}}
    return out or None


def partitioned_conv_unit_rooflines(prof, steps, rows, L, N):
    """The two kernels of the partitioned convolution (N > 8193 taps) against the units that bind them -- HBM is the wrong
    yardstick for the second one: `xspec_kernel` streams (HBM: x in, window spectra out);
    `macinv_pair_kernel` walks 2 nparts + 1 tile-sized operands (69 632 B each) per pair of output tiles through the CUs'
    vector-memory path, which takes 64 B per clock and CU (256 CUs x 64 B x 2.4 GHz), almost all of it L2 hits."""
    def ms_of(part):
        return sum(sum(a.elapsed_time(b) for a, b, _ in recs) for name, recs in prof.items() if part in name) / steps

    nparts = -(-(N - 1) // 8192) if N > 8193 else 1
    tiles = -(-L // 8192)
    out = {}
    # one record covers both kernels ("xspec_kernel+macinv_pair_kernel"): split by the trace's shares is not possible here, so
    # the pair is reported as a whole against the sum of its two models
    ms = ms_of("macinv")
    if ms > 0:
        spectra = rows * tiles * 69632                       # window spectra written by xspec, read back by the walk
        hbm = 4 * rows * L + 2 * spectra + 4 * rows * L      # x in, spectra out and in (once-through), y out
        walk = rows * -(-tiles // 2) * (2 * nparts + 1) * 69632
        l1_peak = 256 * 64 * 2.4e9 / 1e9
        out = {"kernels": "xspec_kernel + macinv_pair_kernel", "ms_per_step": ms,
               "hbm": {"bound": "hbm", "bytes_once_through": hbm, "achieved": hbm / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                       "unit": "GB/s", "frac": hbm / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
               "l1_intake": {"bound": "l1_intake", "bytes_through_the_cus": walk, "achieved": walk / (ms * 1e-3) / 1e9,
                             "peak": l1_peak, "unit": "GB/s", "frac": walk / (ms * 1e-3) / 1e9 / l1_peak,
                             "model": f"{2 * nparts + 1} operands of 69632 B per pair of 8192-sample output tiles"}}
    return out or None


class GpuSampler:
    """Shader clock and board power from the amdgpu hwmon files (readable without privileges), sampled by a thread every
    50 ms while a loop runs.  The box lists every card of the node (other tenants' included): the card is the one whose
    PCI address is the current torch device's (falling back to all cards, reported per card, when that cannot be told)."""

    def __init__(self, period=0.05, device=None):
        import glob
        import threading

        self.period = period
        self.cards = {}
        want = None
        try:
            import torch

            pr = torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else device)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            want = None
        self.matched = False
        found = {}
        for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            f, p = os.path.join(h, "freq1_input"), os.path.join(h, "power1_input")
            if os.path.exists(f):
                addr = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(h))))
                found[h] = (f, p if os.path.exists(p) else None, addr)
        mine = {h: v for h, v in found.items() if want is not None and v[2].lower() == want}
        self.matched = bool(mine)
        self.cards = {h: v[:2] for h, v in (mine or found).items()}
        self.addr = {h: v[2] for h, v in found.items()}
        self.samples = {h: [] for h in self.cards}
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop.is_set():
            for h, (f, p) in self.cards.items():
                self.samples[h].append((self._read(f), self._read(p) if p else None))
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()

    def summary(self):
        best = None
        for h, xs in self.samples.items():
            fr = [a for a, _ in xs if a is not None]
            pw = [b for _, b in xs if b is not None]
            if not fr:
                continue
            rec = {"sclk_mhz_mean": sum(fr) / len(fr) / 1e6, "sclk_mhz_min": min(fr) / 1e6, "sclk_mhz_max": max(fr) / 1e6,
                   "power_w_mean": sum(pw) / len(pw) / 1e6 if pw else None, "samples": len(fr),
                   "source": h + "/freq1_input", "pci": self.addr.get(h),
                   "card": "matched to the torch device by PCI address" if self.matched else
                           "NOT matched (no PCI address): the card that drew the most power -- may be another tenant's"}
            if best is None or (rec["power_w_mean"] or 0) > (best["power_w_mean"] or 0):
                best = rec
        return best or {"sclk_mhz_mean": None, "samples": 0, "source": None}


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and "RANK" in os.environ:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: using WORLD_SIZE", file=sys.stderr)
    if args.dry:
        dev = torch.device("cpu")
        torch.set_num_threads(max(1, min(4, (os.cpu_count() or 1) // max(world, 1))))
    else:
        if args.backend != "nccl":
            raise SystemExit("--backend gloo is the CPU plumbing mode: combine it with --dry")
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "RANK" in os.environ:  # one process per GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # RCCL builds its communicator (device buffers, proxy threads) lazily at the first collectives; do that now,
            # not inside the timed region (its hipMallocs are device-synchronising: seen as 250 ms "steps" with --warmup 1)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            dist.barrier()
            torch.cuda.synchronize()

    if dist is not None:  # one line per rank in the log: which backend carries the collectives and how many ranks it sees
        print(f"[bench] rank {rank}/{world}: torch.distributed backend={dist.get_backend()} "
              f"world_size={dist.get_world_size()} device={dev}", file=sys.stderr, flush=True)

    def sync():
        if not args.dry:
            torch.cuda.synchronize()

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    if args.reference_default_lengths or args.reference_default_args:
        LENS.update(fsm_fir_len=4000, iir_len=16384, ir_len=60000)
    if args.config == "cfg4":
        out = bench_console(args, torch, dist, dev, world, rank, sync, fence)
    else:
        out = bench_processor(args, torch, dist, dev, world, rank, sync, fence)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def _timed_region(args, torch, dist, dev, step, sync, fence, profile):
    """W warm-up steps, then exactly K steps between barrier + synchronize fences; MAX over ranks."""
    from grafx_amd import ops

    y = None
    if dist is not None:
        # line the ranks up BEFORE the warm-up: their start-up skew (seconds) then is not spent idling at the barrier
        # that opens the timed region (a GPU that has idled for milliseconds runs its next large launches 20-30 % slow)
        sync()
        dist.barrier()
    for _ in range(args.warmup):
        y = step()
    fence()
    # the dominant kernel is timed live, inside the timed region, with a pair of stream events around every launch
    # (recorded on the stream the kernel runs on; no synchronisation, ~120 event records per step)
    import contextlib

    with (ops.profiling() if profile else contextlib.nullcontext()) as prof:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step()
        sync()                        # this rank's K steps are done: stop its clock ...
        elapsed = time.perf_counter() - t0
    fence()                           # ... then the closing barrier; the job's time is the MAX over ranks (below)
    per_rank = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, t)
        per_rank = [float(g.item()) for g in gathered]
        elapsed = max(per_rank)
    return y, elapsed, per_rank, prof


def bench_console(args, torch, dist, dev, world, rank, sync, fence):
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    B = args.batch or 256
    L = args.length or 131072
    G = console_graph()
    render_data = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam"))
    smoother = "ballistics" if args.console_variant == "ballistics" else "iir"
    procs = dry_processors() if args.dry else {k: v.to(dev) for k, v in
                                               hip_processors(args.reference_default_args, energy_smoother=smoother).items()}
    torch.manual_seed(1234)  # identical parameters on every rank (a shared mixing console)
    params_cpu = {t: {k: v.detach() for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
    if args.console_variant in ("longpole", "clamp"):
        params_cpu["compressor"]["z_alpha_pre"] = torch.full_like(params_cpu["compressor"]["z_alpha_pre"],
                                                                  6.0 if args.console_variant == "longpole" else 12.0)
    params = {t: {k: v.to(dev) for k, v in d.items()} for t, d in params_cpu.items()}
    torch.manual_seed(1000 + rank)  # each rank renders its own shard of the batch
    x = torch.randn(B, 32, 2, L, device=dev)
    rd_dev = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)
    # which shard each rank holds: a fingerprint of its first samples, gathered (N distinct values = N distinct shards)
    shard_fp = None
    if dist is not None:
        fp = x.reshape(-1)[:256].double().sum().reshape(1)
        every = [torch.zeros_like(fp) for _ in range(dist.get_world_size())]
        dist.all_gather(every, fp)
        shard_fp = [float(e.item()) for e in every]

    captured = []

    def step():
        if captured:
            return captured[0]()[0]
        with torch.no_grad():
            return render_grafx(procs, x, params, rd_dev, parameters_grad=False)[0]

    if not args.dry:
        # prime the caching allocator with the two signal buffers the loop ping-pongs between (the render
        # returns views of its (B, 111, 2, L) buffer, so one stays alive while the next step allocates):
        # a first hipMalloc of ~30 GB on a freshly booted box can take most of a second.
        pool = [torch.empty(B, render_data.num_nodes, 2, L, device=dev) for _ in range(2)]
        del pool
        with torch.no_grad():  # one single-graph render: loads every kernel's code object, builds the per-device tables
            render_grafx(procs, x[:1], params, rd_dev, parameters_grad=False)
        sync()
        if args.capture:
            from grafx_amd.render import CapturedRender

            captured.append(CapturedRender(procs, x, params, rd_dev))

    y, elapsed, per_rank, prof = _timed_region(args, torch, dist, dev, step, sync, fence,
                                               profile=not (args.capture or args.dry))
    assert torch.isfinite(y).all(), "render produced non-finite samples"
    roof = roofline_from_profile(prof, args.steps, elapsed, B, L) if prof else None

    sustained = None
    if world == 1 and not (args.dry or args.no_sustained):
        y = None             # (checked above) release the timed region's buffer: the loop below keeps two of its own alive
        try:
            sustained = sustained_leg(torch, step, sync, elapsed / args.steps * 1e3)
        except Exception as e:
            sustained = {"error": f"{type(e).__name__}: {e}"}

    train = None
    if not (args.no_train or args.capture):
        y = None
        try:
            train = train_leg(args, torch, dist, dev, world, procs, params, x, rd_dev, L, sync, fence)
        except Exception as e:  # the headline line must survive a failure of the secondary measurement
            train = {"error": f"{type(e).__name__}: {e}"}

    secondary = None
    if world == 1 and not (args.no_secondary or args.dry or args.capture or args.reference_default_lengths
                           or args.reference_default_args):
        x = None  # (the closure `step` is done with it) release the 8.6 GB batch before the other workloads allocate
        try:
            secondary = secondary_leg(torch, dev, sync)
        except Exception as e:  # never lose the headline line to a secondary measurement
            secondary = {"error": f"{type(e).__name__}: {e}"}

    if rank != 0:
        return None
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
                            "forward render, upstream default constructor arguments (flashfftconv=True falls back to the native convolve(), "
                            "even lengths: aliasing compatibility path)"
                            if args.reference_default_args else "forward render, reference-exact lengths (even L+N-1)"),
                   "parallelism": f"batch-shard x{world}", "console_variant": args.console_variant,
                   "launch": "one captured HIP graph per step" if args.capture else "eager render loop"},
        "per_gpu_value": B * L * args.steps / elapsed,
        "world_size": world if dist is None else dist.get_world_size(),
        "backend": None if dist is None else dist.get_backend(),
        "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
        "shard_fingerprints": shard_fp,
        "roofline": roof,
    }
    if args.dry:
        out["dry"] = "CPU plumbing check with pass-through processors: launch/sharding/JSON only, the numbers mean nothing"
    # whole-graph view of the same roofline: 285 row transfers per graph (every node reads its inputs and
    # writes its output once).  Not all of them cross HBM any more, and the rate held against the HBM peak is the one over
    # those that do: the `in` stage's 32 reads are the first eq stage's own (it writes the sources through), and of the
    # mix stage's 64 reads of the channel strips (each strip feeds its bus and the send) the fan-out kernel makes 32,
    # the compressor stage's kernel none (gfx_dynamics_fused_mix_f32 sums the buses while it writes the strips); the
    # bus compressors' kernel likewise produces the master sum (4 of the `out` stage's 5 reads).
    graph_bytes = 285 * B * 2 * L * 4
    from grafx_amd import ops as _ops
    fused_mix = _ops.MIX_FUSION and not (args.reference_default_lengths or args.reference_default_args)
    elided_rows = 0 if args.dry else 32 + (64 + 4 if fused_mix else 32)
    moved = graph_bytes - elided_rows * B * 2 * L * 4
    out["graph_roofline"] = {"algorithmic_bytes_per_step": graph_bytes,
                             "row_transfers": 285, "row_transfers_elided": elided_rows,
                             "moved_bytes_per_step": moved,
                             "achieved_GBps": moved / (ms_per_step * 1e-3) / 1e9,
                             "frac_of_hbm_peak": moved / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_equivalent_GBps": graph_bytes / (ms_per_step * 1e-3) / 1e9}
    if sustained is not None:
        out["sustained"] = sustained
    if train is not None:
        out["training"] = train
    if secondary is not None:
        out["secondary"] = secondary
    if world == 1 and not args.no_cpu_baseline and not args.dry:
        out["cpu_baseline"] = cpu_baseline_console(G, render_data, params_cpu, L)
    # the last key of the line (a reader that keeps only the tail of the output still sees every leg's number)
    summary = {"headline_ms": round(ms_per_step, 3), "headline_roofline_frac": None if roof is None else round(roof["frac"], 4)}
    if sustained is not None:
        summary["sustained_ms"] = round(sustained["ms_per_step"], 3)
    if train is not None and "ms_per_step" in train:
        summary["training_ms"] = round(train["ms_per_step"], 3)
        summary["training_peak_GiB"] = None if train.get("peak_mem_GiB") is None else round(train["peak_mem_GiB"], 2)
    for k, v in (secondary or {}).items():
        summary[k + "_ms"] = round(v["ms_per_step"], 3) if "ms_per_step" in v else v.get("error")
    if "cpu_baseline" in out and isinstance(out["cpu_baseline"], dict) and "value" in out["cpu_baseline"]:
        summary["cpu_baseline_samples_per_s"] = round(out["cpu_baseline"]["value"], 1)
    out["summary"] = summary
    return out


def train_leg(args, torch, dist, dev, world, procs, params, x, rd_dev, L, sync, fence):
    """BASELINE configs[4]: shared parameters, batch sharded over the GPUs, forward + backward + one flat gradient
    all-reduce, at --train-batch graphs per GPU.  Secondary measurement; `value` of the line stays the forward render."""
    import torch.nn as nn

    from grafx_amd.parallel import all_reduce_gradients
    from grafx_amd.render import render_grafx

    if not args.dry:
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
    Bt = min(args.train_batch, x.shape[0])
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
    for _ in range(args.train_steps):
        train_step()
    sync()
    dt = time.perf_counter() - t1
    fence()
    grad_sync = None
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # after the all-reduce every rank must hold the same gradients: compare them rank by rank (a few KB)
        flat = torch.cat([p.grad.reshape(-1) for p in plist if p.grad is not None])
        every = [torch.zeros_like(flat) for _ in range(dist.get_world_size())]
        dist.all_gather(every, flat)
        grad_sync = {"ranks": len(every), "grad_abs_sum": float(flat.abs().sum()),
                     "max_abs_diff_across_ranks": float(max((e - every[0]).abs().max() for e in every))}
    return {"what": "forward + backward + flat all-reduce of shared-parameter gradients", "batch_per_gpu": Bt,
            "steps": args.train_steps, "ms_per_step": dt / args.train_steps * 1e3,
            "value": world * Bt * L * args.train_steps / dt, "unit": "audio samples/s",
            "grad_floats": sum(p.grad.numel() for p in plist if p.grad is not None), "grad_sync": grad_sync,
            "peak_mem_GiB": None if args.dry else torch.cuda.max_memory_allocated() / 2**30}


def processor_case(cfg, torch, dev, rank, batch=None, length=None, lens=None):
    """BASELINE configs[1] / configs[2] as (step function, rows, channels, length, description)."""
    from grafx_amd.processors import ParametricEqualizer, STFTMaskedNoiseReverb

    LENS = lens or globals()["LENS"]
    torch.manual_seed(1000 + rank)
    if cfg == "cfg2":
        R, C, L = batch or 1024, 1, length or 480000
        proc = ParametricEqualizer(num_filters=6, processor_channel="mono", flashfftconv=False,
                                   fsm_fir_len=LENS["fsm_fir_len"]).to(dev)
        p = {k: torch.randn(R, 1, 6, device=dev) for k in ("w0", "q_inv", "log_gain")}
        what = (f"BASELINE configs[1]: ParametricEqualizer(num_filters=6, mono, fsm_fir_len={LENS['fsm_fir_len']}) on "
                f"{R} x {C} x {L}, parameters randn (std 1)")
    else:
        R, C, L = batch or 512, 2, length or 240000
        proc = STFTMaskedNoiseReverb(ir_len=LENS["ir_len"], flashfftconv=False).to(dev)
        p = {k: torch.randn(R, 2, 193, device=dev) for k in ("init_log_magnitude", "delta_log_magnitude")}
        what = (f"BASELINE configs[2]: STFTMaskedNoiseReverb(ir_len={LENS['ir_len']}, pseudo_midside) on {R} x {C} x {L}, "
                f"parameters randn (std 1)")
    x = torch.randn(R, C, L, device=dev)

    def step():
        with torch.no_grad():
            return proc(x, **p)

    return step, R, C, L, what


def call_roofline(R, C, L, ms_per_step):
    call_bytes = 8 * R * C * L  # read x once, write y once
    gbps = call_bytes / (ms_per_step * 1e-3) / 1e9
    return {"algorithmic_bytes_per_call": call_bytes, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS}


def console_case(torch, dev, B, L, lens, seed=1234, energy_smoother="iir", z_alpha_pre=None, keep_signal_buffer=True):
    """The headline console graph on a resident batch of B graphs as a step function (forward render).
    ``energy_smoother``: the compressors' envelope follower ("iir" as in the headline, or "ballistics");
    ``z_alpha_pre``: set every compressor's smoother logit to this value instead of randn * 0.1 (6 -> a pole at 0.9975,
    a 400-sample time constant: the one-pole smoother's memory is then thousands of samples, not tens)."""
    from grafx_amd.data import convert_to_tensor
    from grafx_amd.render import prepare_render, render_grafx, reorder_for_fast_render
    from grafx_amd.utils import create_empty_parameters

    G = console_graph()
    procs = {k: v.to(dev) for k, v in hip_processors(lens=lens, energy_smoother=energy_smoother).items()}
    torch.manual_seed(seed)
    params = {t: {k: v.detach().to(dev) for k, v in d.items()} for t, d in create_empty_parameters(procs, G, std=0.1).items()}
    if z_alpha_pre is not None:
        params["compressor"]["z_alpha_pre"] = torch.full_like(params["compressor"]["z_alpha_pre"], float(z_alpha_pre))
    x = torch.randn(B, 32, 2, L, device=dev)
    rd = prepare_render(reorder_for_fast_render(convert_to_tensor(G), method="beam")).to(dev)

    def step():
        with torch.no_grad():
            return render_grafx(procs, x, params, rd, parameters_grad=False, keep_signal_buffer=keep_signal_buffer)[0]

    return step


def secondary_leg(torch, dev, sync, steps=20, warmup=5):
    """The other single-GPU workloads, a few milliseconds to a few hundred each, appended to the headline line so that the
    driver's own run records them: BASELINE configs[1] and configs[2] with the exact (odd) tap counts, and the "compat"
    legs of SURVEY 8d -- the same two processors and the console graph with upstream's default even tap counts
    (4000 / 16384 / 60000), where every convolve() takes the odd-length aliasing path.  Per leg:
    ms per call, the call's algorithmic-bytes roofline and the dominant kernel's."""
    from grafx_amd import ops

    out = {}
    legs = [("cfg2", "cfg2", None, steps, warmup), ("cfg3", "cfg3", None, steps, warmup),
            ("cfg4_longpole", "cfg4v", dict(z_alpha_pre=6.0), 10, 3), ("cfg4_clamp", "cfg4v", dict(z_alpha_pre=12.0), 10, 3),
            ("cfg4_ballistics", "cfg4v", dict(energy_smoother="ballistics"), 10, 3),
            ("cfg4_output_only", "cfg4v", dict(keep_signal_buffer=False), 10, 3),
            ("cfg2_compat", "cfg2", REFERENCE_DEFAULT_LENS, 5, 2), ("cfg3_compat", "cfg3", REFERENCE_DEFAULT_LENS, 5, 2),
            ("cfg4_compat", "cfg4", REFERENCE_DEFAULT_LENS, 5, 2)]
    for key, cfg, lens, n, w in legs:
        torch.cuda.empty_cache()
        moved_bytes, chains = None, None
        try:
            if cfg == "cfg4v":
                # the headline console at the headline size with another compressor setting (SURVEY 8d "a ballistics
                # variant"; the long-pole leg is the compressor's data-dependent path: bench.py's randn * 0.1 logits put
                # every smoother pole near 0.5)
                R, C, L = 256, 2, 131072
                variant, lens = lens, None
                step = console_case(torch, dev, R, L, LENS, **variant)
                what = (f"BASELINE configs[3] console graph at batch {R}, L={L}, the headline's tap counts, "
                        + (f"every compressor's smoother logit z_alpha_pre = {variant['z_alpha_pre']:g} "
                           + ("(pole 0.9975: the look-back tiles)" if variant["z_alpha_pre"] < 8 else
                              "(the clamp, pole 1 - 1e-5: the N-tap truncation term is alive, core/envelope.py:34-60)")
                           if "z_alpha_pre" in variant
                           else "the headline's processors and parameters, render_grafx(keep_signal_buffer=False): the output node "
                                "only -- an EXTENSION of upstream's API (which always returns every node's signal), not the headline: "
                                "rows nothing reads are not written (the sources' copy, the channel strips' and bus compressors' "
                                "outputs, which only feed the fused routing sums)" if "keep_signal_buffer" in variant
                           else "Compressor(energy_smoother='ballistics') (attack / release recursion, z_alpha_pre ~ randn * 0.1)"))
                unit, units = "audio samples/s", R * L
                call_bytes = 285 * R * 2 * L * 4
                # row transfers that never cross HBM: the sources' write-through + the fused routing sums (as in the
                # headline's graph_roofline); the ballistics compressor does not take the routing sum; the output-only
                # render also skips the sources' copy and the rows that only feed the fused sums
                # (at the clamp the compressors' rows leave the tile grid: written by the row kernel, read back by the sums)
                elided = (32 + 32 if "energy_smoother" in variant else 32 if variant.get("z_alpha_pre", 0) >= 8 else 32 + 64 + 4) \
                    + (32 + 32 + 4 if "keep_signal_buffer" in variant else 0)
                moved_bytes = (285 - elided) * R * 2 * L * 4
            elif cfg == "cfg4":
                R, C, L = 256, 2, 131072     # the stated configuration (rounds 4-5 ran a quarter of it)
                step = console_case(torch, dev, R, L, lens)
                what = (f"BASELINE configs[3] console graph at batch {R}, L={L}, upstream default tap counts "
                        f"{lens['fsm_fir_len']} / {lens['iir_len']} / {lens['ir_len']}: every convolve() aliases (odd L + N - 1)")
                unit, units = "audio samples/s", R * L
                call_bytes = 285 * R * 2 * L * 4
                n, w = 3, 3     # (three warm-ups: the caching allocator needs them to hold the 4-5 GB transients of this path)
                chains = [(False, 36 * 2 * R, L + lens["fsm_fir_len"] - 1), (True, 36 * R, L + lens["iir_len"] - 1),
                          (False, 2 * R, L + lens["ir_len"] - 1)]
            else:
                step, R, C, L, what = processor_case(cfg, torch, dev, 0, lens=lens)
                unit, units = "channel-samples/s", R * C * L
                call_bytes = 8 * R * C * L
                if lens is not None:
                    chains = [(False, R * C, L + (lens["fsm_fir_len"] if cfg == "cfg2" else lens["ir_len"]) - 1)]
            for _ in range(w):
                y = step()
            sync()
            with ops.profiling() as prof:
                t0 = time.perf_counter()
                for _ in range(n):
                    y = step()
                sync()
                elapsed = time.perf_counter() - t0
            assert torch.isfinite(y).all(), f"{key}: non-finite samples"
            ms = elapsed / n * 1e3
            gbps = call_bytes / (ms * 1e-3) / 1e9
            if moved_bytes is None:
                croof = {"algorithmic_bytes_per_call": call_bytes, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS}
            else:   # a console render: the fraction is of the bytes that actually cross HBM (never above 1); the 285
                    # row-transfer yardstick of SURVEY 8d is kept beside it as an equivalent rate
                mg = moved_bytes / (ms * 1e-3) / 1e9
                croof = {"algorithmic_bytes_per_call": call_bytes, "moved_bytes_per_call": moved_bytes, "achieved_GBps": mg,
                         "frac_of_hbm_peak": mg / HBM_PEAK_GBS, "algorithmic_equivalent_GBps": gbps}
            out[key] = {"workload": what, "steps": n, "warmup": w, "ms_per_step": ms,
                        "value": units * n / elapsed, "unit": unit, "call_roofline": croof,
                        "roofline": roofline_from_profile(prof, n, elapsed, R, L, lens)}
            if chains:
                out[key]["valu_roofline"] = czt_valu_roofline(prof, n, chains)
            if cfg == "cfg3":
                N3 = (lens or LENS)["ir_len"]
                out[key]["unit_rooflines"] = partitioned_conv_unit_rooflines(prof, n, R * C, L + N3 - 1 if lens else L, N3)
            del step, y
        except Exception as e:  # one leg must not take the others (or the headline) with it
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return out


def sustained_leg(torch, step, sync, ms_per_step, seconds=6.0):
    """The headline step repeated for >= `seconds` (the 20-step timed region is 0.2 s: too short for a power-limited
    kernel to show the clock it holds), with the shader clock and the board power sampled meanwhile."""
    n = max(50, int(seconds * 1e3 / ms_per_step) + 1)
    for _ in range(3):       # settle the allocator (a third buffer-sized hipMalloc here once cost 0.9 s of the 6)
        y = step()
    sync()
    with GpuSampler() as smp:
        t0 = time.perf_counter()
        for _ in range(n):
            y = step()
        sync()
        dt = time.perf_counter() - t0
    del y
    return {"seconds": dt, "steps": n, "ms_per_step": dt / n * 1e3, **smp.summary()}


def bench_processor(args, torch, dist, dev, world, rank, sync, fence):
    """BASELINE configs[1] / configs[2]: one processor call over a resident batch of rows."""
    if args.dry:
        raise SystemExit("--dry only applies to the console graph")
    step, R, C, L, what = processor_case(args.config, torch, dev, rank, args.batch, args.length)
    step()
    sync()
    y, elapsed, per_rank, prof = _timed_region(args, torch, dist, dev, step, sync, fence, profile=True)
    assert torch.isfinite(y).all(), "processor produced non-finite samples"
    roof = roofline_from_profile(prof, args.steps, elapsed, R, L) if prof else None
    if rank != 0:
        return None
    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "channel-samples/sec through one processor call (all GPUs)",
        "value": world * R * C * L * args.steps / elapsed, "unit": "channel-samples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (randn signals and parameters)",
        "config": {"workload": what, "rows_per_gpu": R, "channels": C, "audio_len": L,
                   "parallelism": f"row-shard x{world}"},
        "world_size": world if dist is None else dist.get_world_size(),
        "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
        "roofline": roof,
        "call_roofline": call_roofline(R, C, L, ms_per_step),
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_proc(args.config, L)
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
