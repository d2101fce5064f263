"""Shape helpers for the UNet glue (reference nn/utils.py:7-74).  The QASM / qiskit export
part of that file (:77-129) is hardware-export tooling outside the hot path."""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn.functional as F


def autocrop(x, y):
    """Centre-crop ``y`` to the spatial size of ``x`` (reference nn/utils.py:7-19)."""
    if x.shape > y.shape:
        warnings.warn("x is larger than y. Cropping x to match y")
        return autocrop(y, x)
    (hx, wx), (hy, wy) = x.shape[2:4], y.shape[2:4]
    return x, y[:, :, (hy - hx) // 2: (hy + hx) // 2, (wy - wx) // 2: (wy + wx) // 2]


def autopad(x, y):
    """Zero-pad ``y`` to the spatial size of ``x``; the odd pixel goes left/top
    (reference nn/utils.py:22-39)."""
    if x.shape < y.shape:
        warnings.warn("x is smaller than y. Padding x to match y")
        return autopad(y, x)
    dh, dw = x.shape[2] - y.shape[2], x.shape[3] - y.shape[3]
    pads = (math.ceil(dw / 2), math.floor(dw / 2), math.ceil(dh / 2), math.floor(dh / 2))
    return x, F.pad(y, pads, mode="constant", value=0)


def get_label_embedding(labels: torch.Tensor, width: int, height: int):
    """``0.1 * sin(label + col/20)`` broadcast to (b, 1, width, height)
    (reference nn/utils.py:42-56, the variant bound at :74)."""
    ramp = torch.arange(width, device=labels.device) / 20
    mask = 0.1 * torch.sin(labels.reshape(-1, 1) + ramp.reshape(1, -1))
    return mask.reshape(labels.shape[0], 1, width, 1).expand(-1, 1, width, height)
