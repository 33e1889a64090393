"""Tensor form of a graph (mirrors grafx.data.tensor.GRAFXTensor — reference src/grafx/data/tensor.py:9-103)."""
from dataclasses import dataclass
from typing import Union

import torch

from .configs import NodeConfigs


@dataclass
class GRAFXTensor:
    node_types: torch.LongTensor
    edge_indices: torch.LongTensor
    counter: int
    batch: bool
    config: NodeConfigs
    config_hash: str
    invalid_op: str
    edge_types: Union[torch.LongTensor, None] = None
    rendering_order_method: Union[str, None] = None
    rendering_orders: Union[torch.LongTensor, None] = None
    type_sequence: Union[torch.LongTensor, None] = None

    @property
    def num_nodes(self):
        return len(self.node_types)

    @property
    def num_edges(self):
        # reference tensor.py:86-88 returns len(edge_indices) (== 2 for a [2,E] tensor); kept.
        return len(self.edge_indices)

    def to(self, device):
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                self.__dict__[k] = v.to(device)

    def __str__(self):
        rows = []
        for k, v in self.__dict__.items():
            rows.append(f"\n  {k}={list(v.shape) if isinstance(v, torch.Tensor) else repr(v)}")
        return "GRAFXTensor(" + ", ".join(rows) + "\n)"
