"""The per-type render loop (mirrors grafx.render.graph.render_grafx —
reference src/grafx/render/graph.py:16-177).

This is the *caller* of the hot path: for every scheduled type it gathers the
input rows from the signal buffer, calls ``processors[type](*signals, **params)``
(the drop-in boundary, reference graph.py:143-145) and writes the result back.
The loop, the routing and the parameter plumbing stay in Python, exactly as
upstream; the processors are the HIP-backed modules of ``grafx_amd.processors``
(or any ``nn.Module`` with the same interface, e.g. the CPU oracle in tests).
"""
import contextlib
import os
import warnings

import torch

from ..data.configs import UTILITY_TYPES
from .core import (
    aggregate_tensor,
    create_signal_buffer,
    expand_tensor_or_tensor_dict,
    flatten_batch_and_node,
    inplace_write_tensor,
    read_tensor_or_tensor_dict,
)


# Training forward: keep the dynamics stages' smoother scan (R x L floats per stage) for their backward.  Rounds 2-5 kept it
# (the alternative was a pass of its own over every row); since round 6 the backward tiles rebuild the scan from the
# samples they read anyway (gfx_dynamics_bwd_rescan_ws_f32), which takes 4 bytes per sample out of the forward AND the
# backward kernel and 4.8 GB at 256 graphs out of the step's peak: off by default (GRAFX_KEEP_SCAN=1: round 5's path).
KEEP_SMOOTHER_SCAN = os.environ.get("GRAFX_KEEP_SCAN", "0") == "1"
# Where the parameter-only work of the later stages (filter design, the reverb's impulse response and spectra) runs:
#   "under_first"   on a side stream underneath the first processor stage's signal kernel (the convolution of the first
#                   equaliser stage in a console: compute-bound, the side kernels take CUs from it);
#   "under_second"  underneath the second processor stage (the compressors in a console: memory-bound, idle ALUs);
#   "inline"        no side stream: every stage designs its own filters on the main stream right before it runs.
# Measured on the headline graph: profiles/r4/prepare_stream_ab.md.
PREPARE_MODE = "under_first"

# The adjoint of a routing sum whose sources feed the same destinations in blocks (console: eight strips -> their bus + the
# send) stays in block form -- k rows per graph instead of k * m -- when the stage that wrote those rows can read it so
# (see _block_fan, autograd.grad_source).  False: always expand (round 5's path: gather_sum_fanout writes every row).
BLOCK_FAN_ADJOINT = True


def _gather_plan(step, device):
    """(src_idx, seg_ptr, n_out) for gfx_gather_sum_f32, or None when the step is a plain slice read.

    Built once per (step, device) from the reference's own descriptors: ``source_reads[0]`` says which
    buffer rows feed the step, ``aggregations[0]`` how they collapse onto its nodes."""
    cache = step.__dict__.setdefault("_plans", {})
    key = (device.type, device.index)
    if key in cache:
        return cache[key]
    read, agg = step.source_reads[0], step.aggregations[0]
    plan = None
    if read.method == "slice" and agg.method == "none":
        cache[key] = None
        return None
    if read.method == "slice":
        sources = list(range(read.idx[0], read.idx[1]))
    else:
        sources = read.idx.tolist()
    E = len(sources)
    if agg.method == "none":
        seg = list(range(E + 1))
    elif agg.method == "sum":
        seg = [0, E]
    else:
        slots = agg.idx.tolist()
        if any(b < a for a, b in zip(slots, slots[1:])):
            cache[key] = False  # unsorted scatter: leave it to the generic path
            return False
        n_out = max(slots) + 1
        seg = [0] * (n_out + 1)
        for j in slots:
            seg[j + 1] += 1
        for j in range(n_out):
            seg[j + 1] += seg[j]
    n_out = len(seg) - 1
    fan = None
    uniq = sorted(set(sources))
    if n_out <= 8 and len(uniq) < E:  # some source feeds several destinations: read each source once
        masks = {u: 0 for u in uniq}
        for j in range(n_out):
            for e in range(seg[j], seg[j + 1]):
                masks[sources[e]] |= 1 << j
        fan = (torch.tensor(uniq, dtype=torch.long, device=device),
               torch.tensor([masks[u] for u in uniq], dtype=torch.long, device=device))
    plan = (torch.tensor(sources, dtype=torch.long, device=device), torch.tensor(seg, dtype=torch.long, device=device),
            n_out, fan)
    cache[key] = plan
    return plan


def _transposed_plan(step, plan, device):
    """The adjoint of a gather plan: for every distinct source row, the list of destination slots it fed.
    -> (unique source rows (list), dst_idx tensor, seg_ptr tensor, contiguous?, fan) -- fan = (slot indices, per-slot bit
    mask over the unique source rows) when there are at most 32 of them: the adjoint then reads every slot's gradient once
    (gfx_gather_sum_fanout_f32) instead of once per source row."""
    cache = step.__dict__.setdefault("_plans_T", {})
    key = (device.type, device.index)
    if key not in cache:
        src, seg = plan[0].tolist(), plan[1].tolist()
        by_src = {}
        for j in range(len(seg) - 1):
            for e in range(seg[j], seg[j + 1]):
                by_src.setdefault(src[e], []).append(j)
        uniq = sorted(by_src)
        dst, ptr = [], [0]
        for u in uniq:
            dst.extend(by_src[u])
            ptr.append(len(dst))
        fan = None
        if len(uniq) <= 32:
            masks = {}
            for k, u in enumerate(uniq):
                for j in by_src[u]:
                    masks[j] = masks.get(j, 0) | (1 << k)
            slots = sorted(masks)
            fan = (torch.tensor(slots, dtype=torch.long, device=device),
                   torch.tensor([masks[j] for j in slots], dtype=torch.long, device=device))
        cache[key] = (uniq, torch.tensor(dst, dtype=torch.long, device=device),
                      torch.tensor(ptr, dtype=torch.long, device=device),
                      uniq == list(range(uniq[0], uniq[0] + len(uniq))), fan)
    return cache[key]


def _block_fan(step, plan, device):
    """Block structure of a gather plan's adjoint: when the distinct source rows are contiguous and fall into k blocks of m
    >= 2 consecutive rows that feed the SAME destination slots (the eight channel strips of a console bus: their bus and the
    send), the adjoint has only k distinct rows per graph -> (first source row, k, m, slot index tensor, segment pointer
    tensor) for gfx_gather_sum_f32 over the destination gradients; else None.  The largest such m is taken."""
    cache = step.__dict__.setdefault("_block_fan", {})
    key = (device.type, device.index)
    if key not in cache:
        cache[key] = None
        src, seg = plan[0].tolist(), plan[1].tolist()
        by_src = {}
        for j in range(len(seg) - 1):
            for e in range(seg[j], seg[j + 1]):
                by_src.setdefault(src[e], []).append(j)
        uniq = sorted(by_src)
        n = len(uniq)
        if n >= 2 and uniq == list(range(uniq[0], uniq[0] + n)):
            dests = [tuple(by_src[u]) for u in uniq]
            for m in range(n, 1, -1):
                if n % m == 0 and all(dests[i] == dests[i - i % m] for i in range(n)):
                    idx, ptr = [], [0]
                    for blk in range(n // m):
                        idx.extend(dests[blk * m])
                        ptr.append(len(idx))
                    cache[key] = (uniq[0], n // m, m, torch.tensor(idx, dtype=torch.long, device=device),
                                  torch.tensor(ptr, dtype=torch.long, device=device))
                    break
    return cache[key]


def _mix_schedule(step, nxt, device):
    """When `nxt` is a routing-sum stage that adds up rows of `step` (and possibly finished rows of other stages):
    {"sched", "n_acc", "extras", "n_pre"} with which a processor that ``accepts_mix`` computes the sums itself
    (ops.mix_schedule: every destination adds its rows in increasing order, as the gather-sum kernels do); else None."""
    from .. import ops

    cache = step.__dict__.setdefault("_mix_sched", {})
    key = (device.type, device.index, id(nxt))
    if key not in cache:
        cache[key] = None
        plan = _gather_plan(nxt, device)
        d0, d1 = step.dest_write.idx
        e0 = nxt.dest_write.idx[0]
        if plan:
            src, seg = plan[0].tolist(), plan[1].tolist()
            sched = ops.mix_schedule([[v - d0 for v in src[seg[j]:seg[j + 1]]] for j in range(plan[2])], d1 - d0)
            if sched is not None:
                codes, n_acc, pre, post = sched
                extras = [(d0 + r - e0, c) for r, c in pre + post]
                cache[key] = {"sched": torch.tensor(codes, dtype=torch.long, device=device), "n_acc": n_acc,
                              "extras": torch.tensor(extras, dtype=torch.long, device=device) if extras else None,
                              "n_pre": len(pre), "extra_rows": [d0 + r for r, _ in pre + post]}
    return cache[key]


def _reads_rows(step, a, b):
    """Does the stage read a buffer row in [a, b)?  Cached on the stage like _touches_inputs: an index read lives on the
    device, and asking it costs a host sync per render (illegal while the render is captured into a HIP graph)."""
    read = step.source_reads[0]
    if read.method == "slice":
        return read.idx[0] < b and a < read.idx[1]
    cache = step.__dict__.setdefault("_reads_rows", {})
    if (a, b) not in cache:
        rows = read.idx.tolist() if isinstance(read.idx, torch.Tensor) else list(read.idx)
        cache[(a, b)] = any(a <= r < b for r in rows)
    return cache[(a, b)]


def _plan_max_row(step, plan):
    """Highest buffer row a gather plan reads (cached: the plan lives on the device)."""
    cache = step.__dict__.setdefault("_plan_max", [])
    if not cache:
        cache.append(int(plan[0].max()))
    return cache[0]


def _gather(ops, buf, plan, out):
    if plan[3] is not None and ops.gather_sum_fanout(buf, plan[3][0], plan[3][1], out):
        return out
    return ops.gather_sum(buf, plan[0], plan[1], out)


_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


_PREPARE_STREAMS = {}


def _prepare_stream(device):
    key = (device.type, device.index)
    if key not in _PREPARE_STREAMS:
        _PREPARE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _PREPARE_STREAMS[key]


def _first_order(render_data):
    return 1


def _tee_range(render_data, processors, n_src, device):
    """Source rows [a, b) that the first stage reads as a plain slice through a tee-capable processor, or None."""
    if render_data.max_order < 1:
        return None
    step = render_data.iter_list[_first_order(render_data)]
    read = step.source_reads[0]
    proc = processors[step.node_type] if step.node_type in processors else None
    if proc is None or not getattr(proc, "accepts_tee", False):
        return None
    if read.method != "slice" or read.idx[1] > n_src or _gather_plan(step, device) is not None:
        return None
    return tuple(read.idx)


def _complement(rng, n):
    if rng is None:
        return [(0, n)]
    a, b = rng
    return [(lo, hi) for lo, hi in ((0, a), (b, n)) if hi > lo]


def _touches_inputs(read, n_src):
    """Does this read access a source row?  Cached on the descriptor: an index read lives on the device, and asking
    it costs a host sync (which would also be illegal while the render is being captured into a HIP graph)."""
    cache = read.__dict__.setdefault("_touches", {})
    if n_src not in cache:
        if read.method == "slice":
            cache[n_src] = read.idx[0] < n_src
        elif read.method == "index":
            idx = read.idx
            cache[n_src] = bool((idx < n_src).any()) if isinstance(idx, torch.Tensor) else any(i < n_src for i in idx)
        else:
            cache[n_src] = False
    return cache[n_src]


def _any_requires_grad(p):
    if isinstance(p, torch.Tensor):
        return p.requires_grad
    return any(_any_requires_grad(v) for v in p.values()) if hasattr(p, "values") else False


def _wants_grad(input_signals, per_type_parameters, common_parameters):
    return torch.is_grad_enabled() and (input_signals.requires_grad or _any_requires_grad(per_type_parameters)
                                        or (common_parameters is not None and _any_requires_grad(common_parameters)))


class GenericRenderPathWarning(UserWarning):
    """A render of CUDA signals that does not take the in-place buffer path (see _buffer_io_reason)."""


def _buffer_io_reason(processors, input_signals, render_data, per_type_parameters):
    """None when the render can take the in-place buffer path (gradients are handled by _BufferRenderFn around it), else
    what keeps it off: the render then runs upstream's loop (render/graph.py:104-175 of the reference: copies on read,
    torch routing, one processor call per stage -- the processors themselves still run their HIP kernels)."""
    if not input_signals.is_cuda:
        return "the signals are not on a GPU"
    if render_data.method == "one-by-one":
        return "the schedule is 'one-by-one' (no type batching)"
    if not render_data.siso_only:
        return "the graph holds multi-input / multi-output processors (render/prepare.py:109-192 of the reference)"
    for step in render_data.iter_list[1:]:
        if step.node_type in processors:
            if not hasattr(processors[step.node_type], "render_into"):
                return f"processor type {step.node_type!r} ({type(processors[step.node_type]).__name__}) has no render_into()"
        elif step.node_type not in UTILITY_TYPES:
            return f"node type {step.node_type!r} has no processor"
        if step.dest_write.method != "slice" or len(step.source_reads) != 1 or step.source_reads[0].method == "none":
            return f"stage {step.node_type!r} does not read one input and write a contiguous range of rows"
        if _gather_plan(step, input_signals.device) is False:
            return f"stage {step.node_type!r} aggregates through an unsorted scatter"
    return None


def _buffer_io_ok(processors, input_signals, render_data, per_type_parameters):
    """Structural conditions of the in-place buffer path."""
    reason = _buffer_io_reason(processors, input_signals, render_data, per_type_parameters)
    if reason is not None and input_signals.is_cuda:
        # not silent: a CUDA render off the fast path says so (once per reason and call site)
        warnings.warn(f"render_grafx: taking the generic loop instead of the in-place buffer render because {reason}",
                      GenericRenderPathWarning, stacklevel=3)
    return reason is None


def _render_buffer_io(processors, input_signals, per_type_parameters, render_data, common_parameters, aux=None,
                      keep_signal_buffer=True):
    """render_grafx for HIP processors: every stage reads and writes the (B, V, C, L) signal buffer in place
    (no clone / index_select / reshape copies), routing sums run as one gather-sum kernel.
    ``keep_signal_buffer=False`` (an output-only render; the third return value is None): rows that nothing reads are
    not written -- the sources are not copied into the buffer unless a stage reads them from there, and a stage whose
    rows only feed the routing sum fused into its kernel does not store them.
    ``aux``: a dict (training path) in which processors with ``accepts_aux`` keep per-stage by-products of the forward
    pass that their backward needs (key: the stage's order); the stage-wise backward hands it back to them."""
    from .. import ops

    squeeze = input_signals.ndim == 3
    x = input_signals.unsqueeze(0) if squeeze else input_signals
    B, n_src, C, L = x.shape
    # 4-D input: the parameters are per node and shared by the batch (upstream expands them B times,
    # render/graph.py:68-75).  Processors that understand sharing get the un-expanded rows, build their filters
    # once per node and let every batch row read them; the others get the expanded copies.
    shared_tree = per_type_parameters if (not squeeze and common_parameters is None) else None
    if not squeeze:
        expanded_tree = None  # built on first use
        if common_parameters is not None:
            common_parameters = expand_tensor_or_tensor_dict(common_parameters, expand=B, dim=0)
    node_dim = 0 if squeeze else 1
    postprocess = None if squeeze else flatten_batch_and_node

    buf = torch.empty(B, render_data.num_nodes, C, L, device=x.device)
    # The sources must end up in the buffer's first slots (the buffer is returned with every node's signal),
    # but nothing has to wait for that copy: stages that read source rows read them from `x` itself, and the
    # copy runs on a side stream underneath the first (compute-bound) stages.
    # ... and a stage whose processor can "tee" (write its input through to a second destination from the
    # registers that hold it anyway) makes the copy of the rows it reads free.
    lean = not keep_signal_buffer
    teed = None if lean else _tee_range(render_data, processors, n_src, x.device)
    main = torch.cuda.current_stream(x.device)
    rest = [] if lean else _complement(teed, n_src)
    sources_in_buf = not lean          # lean: copied on demand (ensure_sources), on the main stream
    side = _side_stream(x.device) if rest else None
    if side is not None:
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for a, b in rest:
                buf[:, a:b].copy_(x[:, a:b], non_blocking=True)
        # No record_stream() on x / buf: the main stream joins the side stream before this function returns, so
        # everything the caller (or the allocator, on reuse) does with them afterwards is ordered behind the copy.
        # record_stream would instead make the caching allocator hold the 30 GB buffer back until it has *observed*
        # the side stream's event; with the host running a few steps ahead it then cannot recycle the buffer and
        # falls back to a fresh hipMalloc per step (seen as intermittent 150-800 ms steps).
    copied = False  # has the main stream joined the copy yet?
    out_view = None

    def ensure_sources():
        """Output-only render: a stage is about to read source rows from the buffer (a gathered read, a routing sum that
        takes sources) -- put them there now."""
        nonlocal sources_in_buf
        if not sources_in_buf:
            buf[:, :n_src].copy_(x)
            sources_in_buf = True

    def stage_parameters(step, proc):
        nonlocal expanded_tree
        extra = {}
        if squeeze:
            params = read_tensor_or_tensor_dict(per_type_parameters[step.node_type], step.parameter_read, dim=0)
        elif shared_tree is not None and getattr(proc, "accepts_shared_params", False):
            params = read_tensor_or_tensor_dict(shared_tree[step.node_type], step.parameter_read, dim=0)
            extra["_shared_rows"] = step.dest_write.idx[1] - step.dest_write.idx[0]
        else:
            if expanded_tree is None:
                expanded_tree = expand_tensor_or_tensor_dict(per_type_parameters, expand=B, dim=0)
            params = read_tensor_or_tensor_dict(expanded_tree[step.node_type], step.parameter_read, dim=1,
                                                postprocess=flatten_batch_and_node)
        common_i = {}
        if common_parameters is not None:
            common_i = read_tensor_or_tensor_dict(common_parameters, step.dest_write, dim=node_dim,
                                                  postprocess=postprocess)
        return extra, params, common_i

    def prepare_later_stages(after):
        """Parameter-only work of the stages after `after` (filter design, impulse responses, spectra) on a side
        stream, under the signal kernels of the earlier stages; -> {order: (Prepared, event)}."""
        todo = [j for j in range(after + 1, render_data.max_order + 1)
                if hasattr(processors[render_data.iter_list[j].node_type] if render_data.iter_list[j].node_type in processors
                           else None, "prepare")]
        if not todo:
            return {}
        # parameter views (and, where a processor needs them, the batch-expanded copies) are made on the main stream
        args = {j: stage_parameters(render_data.iter_list[j], processors[render_data.iter_list[j].node_type]) for j in todo}
        prep = _prepare_stream(x.device)
        prep.wait_stream(main)  # the parameters may have been produced on the caller's stream
        ready = {}
        with torch.cuda.stream(prep):
            for j in todo:
                proc_j = processors[render_data.iter_list[j].node_type]
                extra_j, params_j, common_j = args[j]
                state = proc_j.prepare(**extra_j, **params_j, **common_j)
                if state is None:
                    continue
                for tns in state.tensors:  # allocated on the side stream, read on the main one
                    tns.record_stream(main)
                event = torch.cuda.Event()
                event.record(prep)
                ready[j] = (state, event)
        return ready

    prepared = None
    launched = 0  # processor stages launched so far
    done = set()  # stages already produced out of schedule order (see below)

    def run_stage(i, mix_with=None):
        """Stage i; `mix_with`: the routing-sum stage the processor is offered to produce too -> did it?"""
        nonlocal copied, prepared, launched
        step = render_data.iter_list[i]
        d0, d1 = step.dest_write.idx
        out_v = buf.narrow(1, d0, d1 - d0)
        plan = _gather_plan(step, x.device)
        node_type = step.node_type
        src_read = step.source_reads[0]
        from_inputs = plan is None and src_read.idx[1] <= n_src  # plain slice of source rows
        if side is not None and not from_inputs and not copied and _touches_inputs(src_read, n_src):
            main.wait_stream(side)
            copied = True
        if lean and not from_inputs and _touches_inputs(src_read, n_src):
            ensure_sources()
        if node_type not in processors:  # in / out / mix: the (summed) input is the output
            if plan is None:
                a, b = step.source_reads[0].idx
                out_v.copy_((x if from_inputs else buf).narrow(1, a, b - a))
            else:
                _gather(ops, buf, plan, out_v)
            return False
        if plan is None:
            a, b = step.source_reads[0].idx
            x_view = (x if from_inputs else buf).narrow(1, a, b - a)
        else:
            x_view = _gather(ops, buf, plan, torch.empty(B, plan[2], C, L, device=x.device))
        proc = processors[node_type]
        extra, params, common_i = stage_parameters(step, proc)
        if teed is not None and i == _first_order(render_data):
            a, b = teed
            extra["tee"] = buf.narrow(1, a, b - a)
        if prepared is not None and i in prepared:
            state, event = prepared[i]
            main.wait_event(event)
            extra["_prepared"] = state
        if (aux is not None and plan is None and KEEP_SMOOTHER_SCAN and getattr(proc, "accepts_aux", False)
                and type(proc) in _tape_safe_types()):
            # only for the exact library types whose backward consumes it (see `trusted` in the backward); a gathered
            # input is a temporary: the backward re-gathers it, same values
            extra["_aux"] = (aux, i)
        mix = None
        if mix_with is not None:
            nxt = render_data.iter_list[mix_with]
            sched = _mix_schedule(step, nxt, x.device)
            if side is not None and not copied and any(r < n_src for r in sched["extra_rows"]):
                # the sum also takes SOURCE rows, which the kernel reads from `buf`: they are filled by the side stream's
                # copy, and the skipped mix stage is the one that would have joined it (a stage reading `x` directly has not)
                main.wait_stream(side)
                copied = True
            if lean and any(r < n_src for r in sched["extra_rows"]):
                ensure_sources()
            e0, e1 = nxt.dest_write.idx
            mix = extra["_mix"] = {"sched": sched["sched"], "n_acc": sched["n_acc"], "extras": sched["extras"],
                                   "n_pre": sched["n_pre"], "out": buf.narrow(1, e0, e1 - e0)}
            # output-only render: rows that only the fused sum reads (not the last stage's, not read by any later stage
            # other than the sum itself) are not stored
            if lean and i != render_data.max_order and not any(
                    _reads_rows(render_data.iter_list[k], d0, d1) for k in range(i + 1, render_data.max_order + 1)
                    if k != mix_with and k not in done):
                mix["skip_rows"] = True
        proc.render_into(x_view, out_v, **extra, **params, **common_i)
        launched += 1
        if prepared is None and PREPARE_MODE != "inline" and launched == (2 if PREPARE_MODE == "under_second" else 1):
            prepared = prepare_later_stages(i)  # this stage is on its way: now design the later ones underneath it
        return mix is not None and bool(mix.get("done"))

    def mix_candidate(i):
        """The routing-sum stage that stage i may produce itself, and the stages to run before stage i for that:
        the stage right behind it -- or the one behind ONE processor stage that does not read stage i's rows (the
        console: the bus compressors, then the reverb, then the master sum of both), which then runs first."""
        step = render_data.iter_list[i]
        proc = processors[step.node_type]
        if not (getattr(proc, "accepts_mix", False) and type(proc) in _mix_safe_types()):
            return None, []
        first = []
        j = i + 1
        if j <= render_data.max_order and render_data.iter_list[j].node_type in processors and j not in done:
            if _reads_rows(render_data.iter_list[j], *step.dest_write.idx):
                return None, []
            first, j = [j], j + 1
        if j > render_data.max_order or render_data.iter_list[j].node_type in processors:
            return None, []
        sched = _mix_schedule(step, render_data.iter_list[j], x.device)
        if sched is None:
            return None, []
        if first:
            f0, f1 = render_data.iter_list[first[0]].dest_write.idx
            if not any(f0 <= r < f1 for r in sched["extra_rows"]):
                return None, []      # the sum does not need the stage in between: keep the schedule's order
        rows_ready = lambda r: r < step.dest_write.idx[0] or any(  # noqa: E731
            render_data.iter_list[f].dest_write.idx[0] <= r < render_data.iter_list[f].dest_write.idx[1] for f in first)
        if not all(rows_ready(r) for r in sched["extra_rows"]):
            return None, []
        return j, first

    for i in range(1, render_data.max_order + 1):
        step = render_data.iter_list[i]
        d0, d1 = step.dest_write.idx
        out_view = buf.narrow(1, d0, d1 - d0)
        if i in done:
            continue
        j, first = mix_candidate(i) if step.node_type in processors else (None, [])
        if j is not None and first and not (ops.MIX_FUSION and L % 4 == 0):
            j, first = None, []      # (the fused kernel would decline: do not reorder for nothing)
        for f in first:
            run_stage(f)
            done.add(f)
        if run_stage(i, mix_with=j):
            done.add(j)
        # (declined: the stage in between has run early and the sum runs at its own place -- still a valid order)
    if side is not None and not copied:
        main.wait_stream(side)  # the returned buffer is complete on the caller's stream
    if lean:
        return (out_view[0] if squeeze else out_view), [], None
    if squeeze:
        return out_view[0], [], buf[0]
    return out_view, [], buf


# ---- training: the same in-place forward, with a stage-wise backward ------------------------------------
def _tape_safe_types():
    """Exact processor classes whose output is linear in their one native autograd node (see the stage-wise backward)."""
    from .. import processors as P

    return (P.ParametricEqualizer, P.Compressor, P.NoiseGate, P.STFTMaskedNoiseReverb, P.BiquadFilter)


def _mix_safe_types():
    """Exact processor classes whose render_into(..., _mix=) writes exactly the stage's output rows and their sums (a user
    subclass may post-process them: it gets the two stages one after the other)."""
    from .. import processors as P

    return (P.Compressor, P.NoiseGate, P.StereoGain)


def _flatten_tree(tree, leaves):
    """Nested dict of tensors -> spec with leaf indices (tensors appended to `leaves`)."""
    if isinstance(tree, torch.Tensor):
        leaves.append(tree)
        return len(leaves) - 1
    if hasattr(tree, "items"):
        return {k: _flatten_tree(v, leaves) for k, v in tree.items()}
    return ("const", tree)


def _unflatten_tree(spec, leaves):
    if isinstance(spec, int):
        return leaves[spec]
    if isinstance(spec, dict):
        return {k: _unflatten_tree(v, leaves) for k, v in spec.items()}
    return spec[1]


class _BufferRenderFn(torch.autograd.Function):
    """render_grafx as ONE autograd node.

    Forward is the in-place buffer render (the inference path, run without a tape).  The signal buffer it returns
    holds every node's output, i.e. every activation the backward needs, so the backward walks the schedule in
    reverse and, per stage, re-evaluates that stage alone on its (detached) input rows with a local tape,
    back-propagates the stage's slice of the buffer gradient through it, and adds the input gradient onto the
    rows the stage read.  Compared with taping the upstream loop (clone-on-read + in-place slice writes into
    one (B, V, C, L) tensor) this never copies or zero-fills the whole buffer gradient per stage — at the console
    graph that was most of the step — and keeps peak memory at two buffers plus one stage's tape."""

    @staticmethod
    def forward(ctx, meta, input_signals, *leaves):
        processors, render_data, p_spec, c_spec = meta
        params = _unflatten_tree(p_spec, leaves)
        common = None if c_spec is None else _unflatten_tree(c_spec, leaves)
        ctx.aux = {}
        with torch.no_grad():
            _, _, buf = _render_buffer_io(processors, input_signals, params, render_data, common, aux=ctx.aux)
        ctx.meta = meta
        # the backward re-traces the stages on the autograd engine's worker thread, which does not see the caller's
        # context-local set_exact_convolution(): carry the setting the forward ran under
        from ..processors.core.convolution import exact_convolution

        ctx.exact = exact_convolution()
        ctx.squeeze = input_signals.ndim == 3
        ctx.n_src = input_signals.shape[0 if ctx.squeeze else 1]
        ctx.save_for_backward(buf, *leaves)
        # The output rows are returned as an output of their own (a small copy) next to the full buffer: a loss that
        # only looks at the output then sends back a small gradient instead of a zero-filled buffer-sized one.
        d0, d1 = render_data.iter_list[render_data.max_order].dest_write.idx
        ctx.out_rows = (d0, d1)
        ctx.set_materialize_grads(False)
        return buf.narrow(0 if ctx.squeeze else 1, d0, d1 - d0).clone(), buf

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out_rows, g_buf):
        from ..processors.core.convolution import exact_convolution_scope

        with exact_convolution_scope(ctx.exact):
            return _BufferRenderFn._backward(ctx, g_out_rows, g_buf)

    @staticmethod
    def _backward(ctx, g_out_rows, g_buf):
        from .. import autograd as diff
        from .. import ops

        processors, render_data, p_spec, c_spec = ctx.meta
        buf, *leaves = ctx.saved_tensors
        squeeze = ctx.squeeze
        if squeeze:
            buf = buf.unsqueeze(0)
            g_out_rows = None if g_out_rows is None else g_out_rows.unsqueeze(0)
            g_buf = None if g_buf is None else g_buf.unsqueeze(0)
        B, V, C, L = buf.shape
        dev = buf.device
        # Gradient of every node's signal, accumulated while walking the schedule backwards.  Held per PART -- the sources
        # and every stage's output rows are one part each, allocated when something first contributes to them and dropped as
        # soon as their stage has been back-propagated (round 6; one buffer-sized tensor before: 28 GiB at the headline batch,
        # of which the sources' and the block-form rows' 16 GiB were never used).  Never zero-filled as a whole: `written`
        # tracks which rows hold a value, the first contribution to a row is a copy, later ones add.
        edges = sorted({(0, ctx.n_src)} | {tuple(render_data.iter_list[j].dest_write.idx)
                                           for j in range(1, render_data.max_order + 1)})
        parts = {}
        written = [False] * V

        def part_view(a, b):  # rows [a, b) when they lie inside one part (allocated on demand), else None
            for pa, pb in edges:
                if pa <= a and b <= pb:
                    t = parts.get((pa, pb))
                    if t is None:
                        t = parts[(pa, pb)] = torch.empty(B, pb - pa, C, L, dtype=buf.dtype, device=dev)
                    return t.narrow(1, a - pa, b - a)
            return None

        def span_view(a, b):  # rows [a, b) as ONE tensor: inside a part, or over whole parts none of which exists yet
            v = part_view(a, b) if len(pieces(a, b)) == 1 else None
            if v is None:
                cover = [e for e in edges if e[0] < b and a < e[1]]
                if cover[0][0] == a and cover[-1][1] == b and not any(e in parts for e in cover):
                    v = torch.empty(B, b - a, C, L, dtype=buf.dtype, device=dev)
                    for pa, pb in cover:       # (the parts are views of it: it lives until the last of them is dropped)
                        parts[(pa, pb)] = v.narrow(1, pa - a, pb - pa)
            return v

        def pieces(a, b):  # [a, b) cut at the part boundaries
            cuts = [(max(a, pa), min(b, pb)) for pa, pb in edges if pa < b and a < pb]
            if not cuts or cuts[0][0] != a or cuts[-1][1] != b or any(x[1] != y[0] for x, y in zip(cuts, cuts[1:])):
                raise RuntimeError(f"render backward: rows [{a}, {b}) are not covered by the schedule's write ranges")
            return cuts

        # Rows whose gradient exists only in block form: (a, b) -> (k distinct rows per graph (B, k, C, L), block size m),
        # row a + i stands for distinct row i // m (see _block_fan).  A stage that reads its output gradient through a row
        # map takes them as they are (autograd.grad_source); anything else gets them written out first.
        virtual = {}

        def materialise(a, b):
            for va, vb in [r for r in virtual if r[0] < b and a < r[1]]:
                rows, m = virtual.pop((va, vb))
                part_view(va, vb).view(B, rows.shape[1], m, C, L).copy_(rows.unsqueeze(2))

        def accumulate(a, b, g):  # rows [a, b) += g  (g: (B, b-a, C, L))
            materialise(a, b)
            for pa, pb in pieces(a, b):
                i = pa
                while i < pb:
                    j = i
                    while j < pb and written[j] == written[i]:
                        j += 1
                    dst, src = part_view(i, j), g.narrow(1, i - a, j - i)
                    if src.data_ptr() == dst.data_ptr():
                        pass  # the stage wrote its input gradient straight into these rows (autograd.GRAD_SINK)
                    elif written[i]:
                        dst.add_(src)
                    else:
                        dst.copy_(src)
                    written[i:j] = [True] * (j - i)
                    i = j

        def settled(a, b):  # rows [a, b) as they stand; rows nothing contributed to are zero
            materialise(a, b)
            for i in range(a, b):
                if not written[i]:
                    part_view(i, i + 1).zero_()
                    written[i] = True
            whole = part_view(a, b)
            return whole if whole is not None else torch.cat([part_view(x, y) for x, y in pieces(a, b)], 1)

        if g_buf is not None:
            for pa, pb in edges:
                part_view(pa, pb).copy_(g_buf.narrow(1, pa, pb - pa))
            written = [True] * V
        if g_out_rows is not None:
            accumulate(*ctx.out_rows, g_out_rows)
        node_dim = 0 if squeeze else 1
        postprocess = None if squeeze else flatten_batch_and_node
        leaf_grads = [None] * len(leaves)
        live = [i for i, t in enumerate(leaves) if t.requires_grad]

        done = None
        for i in range(render_data.max_order, 0, -1):
            if done is not None:
                parts.pop(done, None)  # the previous stage has been back-propagated: nobody reads its output gradient again
                done = None
            step = render_data.iter_list[i]
            d0, d1 = step.dest_write.idx
            if not any(written[d0:d1]):
                continue  # nothing downstream depends on this stage
            done = (d0, d1)
            plan = _gather_plan(step, dev)
            node_type = step.node_type
            trusted = node_type in processors and type(processors[node_type]) in _tape_safe_types()
            # block-form rows stay as they are for a stage that reads its output gradient through a row map
            blocks = virtual.get((d0, d1)) if trusted and getattr(processors[node_type], "reads_grad_source", None) else None
            if blocks is not None and not processors[node_type].reads_grad_source(L):
                blocks = None
            g_out = None if blocks is not None else settled(d0, d1)
            if node_type in processors:
                if plan is None:
                    a, b = step.source_reads[0].idx
                    x_in = buf.narrow(1, a, b - a)
                else:
                    x_in = _gather(ops, buf, plan, torch.empty(B, plan[2], C, L, device=dev))
                with torch.enable_grad():
                    if not getattr(processors[node_type], "accepts_strided_rows", False):
                        x_in = x_in.reshape(-1, C, L)  # the (R, C, L) rows of the upstream contract (a copy)
                    # a stage fed by the sources alone needs no input gradient unless the caller asked for g_x
                    want_gx = ctx.needs_input_grad[1] or (
                        step.source_reads[0].idx[1] > ctx.n_src if plan is None else _plan_max_row(step, plan) >= ctx.n_src)
                    x_in = x_in.detach().requires_grad_(want_gx)  # else: (B, n, C, L) view of the buffer, no copy
                    local = [t.detach().requires_grad_(t.requires_grad) for t in leaves]
                    params = _unflatten_tree(p_spec, local)[node_type]
                    common = None if c_spec is None else _unflatten_tree(c_spec, local)
                    extra = {}
                    if not squeeze and common is None and getattr(processors[node_type], "accepts_shared_params", False):
                        # per-node parameters stay un-expanded: the processor's front-end runs once per node and
                        # its gradient is summed over the batch inside the convolution's backward
                        params = read_tensor_or_tensor_dict(params, step.parameter_read, dim=0)
                        extra["_shared_rows"] = d1 - d0
                    else:
                        if not squeeze:
                            params = expand_tensor_or_tensor_dict(params, expand=B, dim=0)
                            if common is not None:
                                common = expand_tensor_or_tensor_dict(common, expand=B, dim=0)
                        params = read_tensor_or_tensor_dict(params, step.parameter_read, dim=node_dim,
                                                            postprocess=postprocess)
                    common_i = {} if common is None else read_tensor_or_tensor_dict(
                        common, step.dest_write, dim=node_dim, postprocess=postprocess)
                    # Both shortcuts below assume that the processor's output is a LINEAR function of the one native
                    # autograd node that consumes the stage's input view: true for the library's own classes, not for
                    # a user subclass that post-processes super().forward(); so they are enabled for the exact types
                    # only (type(), not isinstance()).
                    if trusted and i in ctx.aux and getattr(processors[node_type], "accepts_aux", False):
                        extra["_aux"] = (ctx.aux, i)   # what the forward render kept for this stage
                    with diff.tape_only(trusted):  # only the stage's tape is wanted here, not its output values
                        y = processors[node_type](x_in, **extra, **params, **common_i)
                    y = y[0] if isinstance(y, tuple) else y
                    wrt = ([x_in] if want_gx else []) + [local[j] for j in live]
                    source = contextlib.nullcontext()
                    if blocks is not None:
                        # the stage's one native node reads the k distinct rows per graph through its row map; the engine
                        # carries a placeholder of the output's shape (one element, zero strides)
                        rows, m = blocks
                        grad_out = diff.tape_placeholder(y.shape, dev)
                        source = diff.grad_source(x_in, rows.view(B * rows.shape[1], 1, C, L).expand(-1, m, -1, -1))
                    else:
                        grad_out = g_out if y.shape == g_out.shape else g_out.reshape(y.shape)
                    with source:
                        sink_rows = None
                        if (trusted and want_gx and plan is None and x_in.ndim == 4 and not any(written[a:b]) and not any(
                                r[0] < b and a < r[1] for r in virtual)):
                            sink_rows = span_view(a, b)
                        if sink_rows is not None:
                            # first (usually only) contribution to these rows: let the stage write it in place
                            with diff.grad_sink(x_in, sink_rows) as sink:
                                grads = torch.autograd.grad(y, wrt, grad_outputs=grad_out, allow_unused=True)
                            if sink.writes > 1:
                                raise RuntimeError(f"{type(processors[node_type]).__name__}: {sink.writes} autograd nodes "
                                                   "wrote the stage's input gradient in place (expected one)")
                        else:
                            grads = torch.autograd.grad(y, wrt, grad_outputs=grad_out, allow_unused=True)
                    if blocks is not None:
                        if source.reads != 1:
                            raise RuntimeError(f"{type(processors[node_type]).__name__}: the stage's block-form output "
                                               f"gradient was read by {source.reads} autograd nodes (expected one)")
                        del virtual[(d0, d1)]
                for j, g in zip(live, grads[1:] if want_gx else grads):
                    if g is not None:
                        leaf_grads[j] = g if leaf_grads[j] is None else leaf_grads[j] + g
                g_in = grads[0].reshape(B, -1, C, L) if want_gx else None
                # this stage's tape (and whatever it kept alive) goes now, not when the next stage rebinds the names:
                # otherwise two stages' temporaries overlap at the peak
                del y, grads, wrt, grad_out, x_in, local, params
                if g_in is None:
                    continue
            else:  # in / out / mix: the (summed) input is the output
                g_in = g_out
            # add the stage's input gradient onto the rows it read
            if plan is None:
                a, b = step.source_reads[0].idx
                accumulate(a, b, g_in)
            else:
                # adjoint of the gather-sum: every source row collects the gradients of the slots it fed --
                # the same gather-sum kernel with the transposed plan
                fanb = _block_fan(step, plan, dev) if BLOCK_FAN_ADJOINT else None
                if fanb is not None:
                    # ... provided the stage that wrote exactly these rows can read the block form (else: no point)
                    span = (fanb[0], fanb[0] + fanb[1] * fanb[2])
                    reader = next((render_data.iter_list[j] for j in range(i - 1, 0, -1)
                                   if tuple(render_data.iter_list[j].dest_write.idx) == span), None)
                    proc = processors[reader.node_type] if reader is not None and reader.node_type in processors else None
                    if not (proc is not None and type(proc) in _tape_safe_types() and hasattr(proc, "reads_grad_source")
                            and proc.reads_grad_source(L)):
                        fanb = None
                if fanb is not None and not any(written[fanb[0] : fanb[0] + fanb[1] * fanb[2]]) and not any(
                        r[0] < fanb[0] + fanb[1] * fanb[2] and fanb[0] < r[1] for r in virtual):
                    # k distinct gradient rows per graph instead of k * m expanded ones: written out only if their reader
                    # cannot take them in this form (materialise)
                    u0, k, m, bidx, bptr = fanb
                    g_in = g_in if g_in.stride(-1) == 1 else g_in.contiguous()
                    virtual[(u0, u0 + k * m)] = (ops.gather_sum(g_in, bidx, bptr, torch.empty(B, k, C, L, device=dev)), m)
                    written[u0 : u0 + k * m] = [True] * (k * m)
                    continue
                uniq, dst_idx, ptr, contiguous, fan = _transposed_plan(step, plan, dev)
                g_src = None
                if contiguous and not any(written[uniq[0] : uniq[0] + len(uniq)]):
                    g_src = span_view(uniq[0], uniq[0] + len(uniq))  # first contribution: gather straight into the rows
                if g_src is None:
                    g_src = torch.empty(B, len(uniq), C, L, device=dev)
                g_in = g_in if g_in.stride(-1) == 1 else g_in.contiguous()
                if fan is None or not ops.gather_sum_fanout(g_in, fan[0], fan[1], g_src):
                    g_src = ops.gather_sum(g_in, dst_idx, ptr, g_src)
                if contiguous:
                    accumulate(uniq[0], uniq[0] + len(uniq), g_src)
                else:
                    for k, u in enumerate(uniq):
                        accumulate(u, u + 1, g_src.narrow(1, k, 1))
        if done is not None:
            parts.pop(done, None)
        # parameters of stages nothing downstream depends on: upstream's taped loop hands back zeros for them (their
        # rows are part of the returned buffer), not None -- optimisers treat the two differently
        # (parameters of a type that has no node in the graph never entered upstream's tape: those stay None)
        def leaf_indices(spec, acc):
            if isinstance(spec, int):
                acc.add(spec)
            elif hasattr(spec, "items"):
                for v in spec.values():
                    leaf_indices(v, acc)
            return acc

        scheduled = {render_data.iter_list[i].node_type for i in range(1, render_data.max_order + 1)} & set(processors)
        taped = set()
        for node_type in scheduled:
            if hasattr(p_spec, "items") and node_type in p_spec:
                leaf_indices(p_spec[node_type], taped)
        if scheduled and c_spec is not None:
            leaf_indices(c_spec, taped)
        for j in live:
            if leaf_grads[j] is None and j in taped:
                leaf_grads[j] = torch.zeros_like(leaves[j])
        g_x = None
        if ctx.needs_input_grad[1]:
            g_x = settled(0, ctx.n_src)
            g_x = (g_x[0] if squeeze else g_x).contiguous()
        return (None, g_x, *leaf_grads)


def _render_buffer_io_with_grad(processors, input_signals, per_type_parameters, render_data, common_parameters):
    leaves = []
    p_spec = _flatten_tree(per_type_parameters, leaves)
    c_spec = None if common_parameters is None else _flatten_tree(common_parameters, leaves)
    out, buf = _BufferRenderFn.apply((processors, render_data, p_spec, c_spec), input_signals, *leaves)
    return out, [], buf


def render_grafx(
    processors,
    input_signals,
    per_type_parameters,
    render_data,
    common_parameters=None,
    parameters_grad=True,
    input_signal_grad=False,
    keep_signal_buffer=True,
):
    """``keep_signal_buffer=False`` (an extension; upstream always returns the buffer): an output-only render on the HIP
    path without gradients -- the third return value is None and rows that nothing reads are not written (the sources'
    copy, the rows of a stage that only feed the routing sum fused into its kernel).  The output is the same bits."""
    method = render_data.method
    ndim = input_signals.ndim
    if ndim in (3, 4) and _buffer_io_ok(processors, input_signals, render_data, per_type_parameters):
        if _wants_grad(input_signals, per_type_parameters, common_parameters):
            return _render_buffer_io_with_grad(processors, input_signals, per_type_parameters, render_data,
                                               common_parameters)
        return _render_buffer_io(processors, input_signals, per_type_parameters, render_data, common_parameters,
                                 keep_signal_buffer=keep_signal_buffer)
    if ndim == 3:
        node_dim, postprocess = 0, None
    elif ndim == 4:
        batch_size, _, channels, audio_len = input_signals.shape
        node_dim, postprocess = 1, flatten_batch_and_node
        per_type_parameters = expand_tensor_or_tensor_dict(per_type_parameters, expand=batch_size, dim=0)
        if common_parameters is not None:
            common_parameters = expand_tensor_or_tensor_dict(common_parameters, expand=batch_size, dim=0)
    else:
        raise Exception(
            f"input_signal has shape of {input_signals.shape} ({ndim} ndims), which is not 3 or 4 dims."
        )

    any_grad = parameters_grad or input_signal_grad
    if input_signal_grad:
        signal_buffer = create_signal_buffer(method, render_data.num_nodes, input_signals)
    else:
        with torch.no_grad():
            signal_buffer = create_signal_buffer(method, render_data.num_nodes, input_signals)

    intermediates_list = []
    output_signals = None

    for i in range(1, render_data.max_order + 1):
        step = render_data.iter_list[i]

        inputs = []
        for read, aggregate in zip(step.source_reads, step.aggregations):
            sig = read_tensor_or_tensor_dict(signal_buffer, read, return_copy=any_grad, dim=node_dim)
            sig = aggregate_tensor(sig, aggregate, dim=node_dim)
            if ndim == 4:
                sig = flatten_batch_and_node(sig)
            inputs.append(sig)

        node_type = step.node_type
        if node_type in processors:
            params = read_tensor_or_tensor_dict(
                per_type_parameters[node_type], step.parameter_read, dim=node_dim, postprocess=postprocess
            )
            if common_parameters is not None:
                common_i = read_tensor_or_tensor_dict(
                    common_parameters, step.dest_write, dim=node_dim, postprocess=postprocess
                )
            else:
                common_i = {}
            result = processors[node_type](*inputs, **params, **common_i)  # <- drop-in boundary
            if isinstance(result, tuple):
                output_signals, intermediates = result
                intermediates_list.append(intermediates)
            else:
                output_signals = result
        elif node_type in UTILITY_TYPES:
            output_signals = inputs
        else:
            raise Exception(f"Wrong node type given: {node_type}")

        if isinstance(output_signals, list):
            if len(output_signals) == 1:
                output_signals = output_signals[0]
            else:
                output_signals = torch.stack(output_signals, -3).view(-1, channels, audio_len)

        if ndim == 4:
            output_signals = output_signals.view(batch_size, -1, channels, audio_len)

        inplace_write_tensor(method, signal_buffer, output_signals, step.dest_write, dim=node_dim)

    return output_signals, intermediates_list, signal_buffer
