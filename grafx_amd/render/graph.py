"""The per-type render loop (mirrors grafx.render.graph.render_grafx —
reference src/grafx/render/graph.py:16-177).

This is the *caller* of the hot path: for every scheduled type it gathers the
input rows from the signal buffer, calls ``processors[type](*signals, **params)``
(the drop-in boundary, reference graph.py:143-145) and writes the result back.
The loop, the routing and the parameter plumbing stay in Python, exactly as
upstream; the processors are the HIP-backed modules of ``grafx_amd.processors``
(or any ``nn.Module`` with the same interface, e.g. the CPU oracle in tests).
"""
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
    if read.method == "slice":
        return read.idx[0] < n_src
    if read.method == "index":
        return bool((read.idx < n_src).any()) if isinstance(read.idx, torch.Tensor) else any(i < n_src for i in read.idx)
    return False


def _buffer_io_ok(processors, input_signals, render_data, per_type_parameters):
    if not input_signals.is_cuda or render_data.method == "one-by-one" or not render_data.siso_only:
        return False
    if torch.is_grad_enabled():
        # training: saved views of a buffer that later stages keep writing would trip autograd's version
        # check, so the reference's clone-on-read loop is used instead
        def any_requires_grad(p):
            if isinstance(p, torch.Tensor):
                return p.requires_grad
            return any(any_requires_grad(v) for v in p.values()) if hasattr(p, "values") else False

        if input_signals.requires_grad or any_requires_grad(per_type_parameters):
            return False
    for step in render_data.iter_list[1:]:
        if step.node_type in processors:
            if not hasattr(processors[step.node_type], "render_into"):
                return False
        elif step.node_type not in UTILITY_TYPES:
            return False
        if step.dest_write.method != "slice" or len(step.source_reads) != 1 or step.source_reads[0].method == "none":
            return False
        if _gather_plan(step, input_signals.device) is False:
            return False
    return True


def _render_buffer_io(processors, input_signals, per_type_parameters, render_data, common_parameters):
    """render_grafx for HIP processors: every stage reads and writes the (B, V, C, L) signal buffer in place
    (no clone / index_select / reshape copies), routing sums run as one gather-sum kernel."""
    from .. import ops

    squeeze = input_signals.ndim == 3
    x = input_signals.unsqueeze(0) if squeeze else input_signals
    B, n_src, C, L = x.shape
    if not squeeze:
        per_type_parameters = expand_tensor_or_tensor_dict(per_type_parameters, expand=B, dim=0)
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
    teed = _tee_range(render_data, processors, n_src, x.device)
    main = torch.cuda.current_stream(x.device)
    side = _side_stream(x.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for a, b in _complement(teed, n_src):
            buf[:, a:b].copy_(x[:, a:b], non_blocking=True)
    x.record_stream(side)
    buf.record_stream(side)
    copied = False  # has the main stream joined the copy yet?
    out_view = None
    for i in range(1, render_data.max_order + 1):
        step = render_data.iter_list[i]
        d0, d1 = step.dest_write.idx
        out_view = buf.narrow(1, d0, d1 - d0)
        plan = _gather_plan(step, x.device)
        node_type = step.node_type
        src_read = step.source_reads[0]
        from_inputs = plan is None and src_read.idx[1] <= n_src  # plain slice of source rows
        if not from_inputs and not copied and _touches_inputs(src_read, n_src):
            main.wait_stream(side)
            copied = True
        if node_type not in processors:  # in / out / mix: the (summed) input is the output
            if plan is None:
                a, b = step.source_reads[0].idx
                out_view.copy_((x if from_inputs else buf).narrow(1, a, b - a))
            else:
                _gather(ops, buf, plan, out_view)
            continue
        if plan is None:
            a, b = step.source_reads[0].idx
            x_view = (x if from_inputs else buf).narrow(1, a, b - a)
        else:
            x_view = _gather(ops, buf, plan, torch.empty(B, plan[2], C, L, device=x.device))
        params = read_tensor_or_tensor_dict(per_type_parameters[node_type], step.parameter_read, dim=node_dim,
                                            postprocess=postprocess)
        common_i = {}
        if common_parameters is not None:
            common_i = read_tensor_or_tensor_dict(common_parameters, step.dest_write, dim=node_dim,
                                                  postprocess=postprocess)
        if teed is not None and i == _first_order(render_data):
            a, b = teed
            processors[node_type].render_into(x_view, out_view, tee=buf.narrow(1, a, b - a), **params, **common_i)
        else:
            processors[node_type].render_into(x_view, out_view, **params, **common_i)
    if not copied:
        main.wait_stream(side)  # the returned buffer is complete on the caller's stream
    if squeeze:
        return out_view[0], [], buf[0]
    return out_view, [], buf


def render_grafx(
    processors,
    input_signals,
    per_type_parameters,
    render_data,
    common_parameters=None,
    parameters_grad=True,
    input_signal_grad=False,
):
    method = render_data.method
    ndim = input_signals.ndim
    if ndim in (3, 4) and _buffer_io_ok(processors, input_signals, render_data, per_type_parameters):
        return _render_buffer_io(processors, input_signals, per_type_parameters, render_data, common_parameters)
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
