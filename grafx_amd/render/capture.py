"""A whole render as ONE HIP graph (inference / serving).

``render_grafx`` is a Python loop that launches ~45 kernels on up to three streams.  At serving sizes (one or a few
graphs per call) the GPU finishes each of them in microseconds and the call is bound by the host: interpreter, ctypes
and launch overhead.  Capturing the loop once and replaying it removes all of that; the kernels, their order, the
side-stream overlap and the results are exactly those of the eager call (bit-identical).

    fast = CapturedRender(processors, example_input, parameters, render_data)
    y, _, buf = fast(new_input)              # or fast(new_input, new_parameters)

Inputs are copied into the static tensors the graph was captured on; the outputs are static too (valid until the next
call).  Shapes, devices and the render data are fixed at capture time.  No autograd.
"""
import torch

from .graph import render_grafx


def _clone_tree(tree):
    if isinstance(tree, torch.Tensor):
        return tree.detach().clone()
    if hasattr(tree, "items"):
        return {k: _clone_tree(v) for k, v in tree.items()}
    return tree


def _copy_tree(dst, src):
    if isinstance(dst, torch.Tensor):
        dst.copy_(src)
    elif hasattr(dst, "items"):
        for k in dst:
            _copy_tree(dst[k], src[k])


class CapturedRender:
    def __init__(self, processors, input_signals, per_type_parameters, render_data, common_parameters=None, warmup=2):
        if not input_signals.is_cuda:
            raise ValueError("CapturedRender needs the HIP path (CUDA/HIP input tensors)")
        self.input_signals = input_signals.detach().clone()
        self.parameters = _clone_tree(per_type_parameters)
        self.common_parameters = None if common_parameters is None else _clone_tree(common_parameters)
        args = (processors, self.input_signals, self.parameters, render_data, self.common_parameters)
        current = torch.cuda.current_stream(input_signals.device)
        stream = torch.cuda.Stream(device=input_signals.device)
        stream.wait_stream(current)
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(warmup):  # code objects, per-device tables, allocator pools: everything lazy happens here
                render_grafx(*args, parameters_grad=False)
        current.wait_stream(stream)
        torch.cuda.synchronize(input_signals.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.outputs = render_grafx(*args, parameters_grad=False)

    def __call__(self, input_signals=None, per_type_parameters=None, common_parameters=None):
        if input_signals is not None:
            self.input_signals.copy_(input_signals)
        if per_type_parameters is not None:
            _copy_tree(self.parameters, per_type_parameters)
        if common_parameters is not None:
            _copy_tree(self.common_parameters, common_parameters)
        self.graph.replay()
        return self.outputs
