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
