"""Spatial soft-argmax (mirror of the reference's utils/softargmax.py) on the HIP kernel.

``SoftArgmax2D(normalized_coordinates=False)(x)`` returns [B, C, 2] in (x, y) order; the plane
reduction is ynet_softargmax2d (one workgroup per plane, single pass, online softmax).
Reference: utils/softargmax.py:10-23 (create_meshgrid), 55-81 (forward).
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import ops


def create_meshgrid(x: torch.Tensor, normalized_coordinates: Optional[bool]):
    assert len(x.shape) == 4, x.shape
    _, _, height, width = x.shape
    if normalized_coordinates:
        xs = torch.linspace(-1.0, 1.0, width, device=x.device, dtype=x.dtype)
        ys = torch.linspace(-1.0, 1.0, height, device=x.device, dtype=x.dtype)
    else:
        xs = torch.linspace(0, width - 1, width, device=x.device, dtype=x.dtype)
        ys = torch.linspace(0, height - 1, height, device=x.device, dtype=x.dtype)
    return torch.meshgrid(ys, xs, indexing="ij")  # pos_y, pos_x


class SoftArgmax2D(nn.Module):
    def __init__(self, normalized_coordinates: Optional[bool] = True) -> None:
        super().__init__()
        self.normalized_coordinates = normalized_coordinates
        self.eps = 1e-6

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        if isinstance(input, ops.LazyPredictor):      # logits = conv1x1(x): predictor and read-out in one launch
            out = ops.pred_softargmax(input.x, input.weight, input.bias)
            if self.normalized_coordinates:
                _, _, h, w = input.shape
                out = out * torch.tensor([2.0 / max(w - 1, 1), 2.0 / max(h - 1, 1)], device=out.device) - 1.0
            return out
        if not torch.is_tensor(input):
            raise TypeError("Input input type is not a torch.Tensor. Got {}".format(type(input)))
        if not len(input.shape) == 4:
            raise ValueError("Invalid input shape, we expect BxCxHxW. Got: {}".format(input.shape))
        out = ops.softargmax2d(input)
        if self.normalized_coordinates:
            _, _, h, w = input.shape
            scale = torch.tensor([2.0 / max(w - 1, 1), 2.0 / max(h - 1, 1)], device=out.device)
            out = out * scale - 1.0
        return out
