"""Round-1 kernels are forward-only: refuse (loudly) to build an autograd graph."""
import torch


def forward_only(*tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise NotImplementedError(
            "grafx_amd HIP processors are forward-only in this release (backward kernels are on the roadmap, "
            "see DESIGN.md). Run inference under torch.no_grad() — e.g. `with torch.no_grad(): render_grafx(...)`."
        )
