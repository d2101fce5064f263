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
    if dh == 0 and dw == 0:
        return x, y              # (F.pad with all-zero pads is a full copy of y: 128 MB per up block at batch 2560)
    pads = (math.ceil(dw / 2), math.floor(dw / 2), math.ceil(dh / 2), math.floor(dh / 2))
    return x, F.pad(y, pads, mode="constant", value=0)


def get_label_embedding(labels: torch.Tensor, width: int, height: int):
    """``0.1 * sin(label + col/20)`` broadcast to (b, 1, width, height)
    (reference nn/utils.py:42-56, the variant bound at :74)."""
    ramp = torch.arange(width, device=labels.device) / 20
    mask = 0.1 * torch.sin(labels.reshape(-1, 1) + ramp.reshape(1, -1))
    return mask.reshape(labels.shape[0], 1, width, 1).expand(-1, 1, width, height)


# --- training-time glue -------------------------------------------------------------------------------------------
# torch has no float64 convolution library path on ROCm: F.unfold / 1x1 Conv2d / bilinear backward run one small
# launch per sample (or an atomics scatter).  The three helpers below state the same maths as single strided-view
# copies and matrix products, which autograd differentiates with the same kind of launches.

def unfold_patches(x: torch.Tensor, kernel_size, padding) -> torch.Tensor:
    """``torch.nn.Unfold(kernel_size, padding)`` (stride 1) rearranged to one row per output pixel:
    ``(b, C, H, W) -> (b * H_out * W_out, C * kh * kw)``, column order ``c * kh * kw + i * kw + j`` as Unfold's
    (reference nn/qconv.py:20, :74-78 unfolds, then transposes to this layout)."""
    kh, kw = kernel_size
    ph, pw = padding
    if ph or pw:
        x = F.pad(x, (pw, pw, ph, ph))
    b, c = x.shape[:2]
    win = x.unfold(2, kh, 1).unfold(3, kw, 1)                      # (b, C, H_out, W_out, kh, kw), a view
    return win.permute(0, 2, 3, 1, 4, 5).reshape(b * win.shape[2] * win.shape[3], c * kh * kw)


def pointwise_conv(conv: torch.nn.Conv2d, x: torch.Tensor) -> torch.Tensor:
    """A 1x1, stride-1, ungrouped ``Conv2d`` as one matrix product over the channel axis."""
    w = conv.weight[:, :, 0, 0]
    if w.shape[0] == 1 and w.shape[1] <= 32 and x.is_cuda and x.dtype == torch.float64 and w.dtype == torch.float64 \
            and x.dim() == 4 and x.numel() > 0 and (conv.bias is None or conv.bias.dtype == torch.float64):
        from .. import circuit as _c
        return _c.conv1x1_head(x, conv.weight, conv.bias)      # forward + one-pass backward in HIP
    if w.shape[0] == 1:
        # one output channel (the UNets' head): a weighted channel sum -- its weight gradient is then a plain
        # reduction instead of a (1 x C) product over B*H*W terms, which the BLAS handles badly
        y = (x * w.view(1, -1, 1, 1)).sum(dim=1, keepdim=True)
    else:
        y = torch.einsum("bchw,oc->bohw", x, w)
    return y if conv.bias is None else y + conv.bias.view(1, -1, 1, 1)


def is_pointwise(conv) -> bool:
    return isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.stride == (1, 1) \
        and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1)


_interp_cache = {}


def _interp_matrix(size: int, device, dtype) -> torch.Tensor:
    """(2*size, size) matrix of the x2 bilinear interpolation (align_corners=False) along one axis, taken from
    torch's own operator applied to the identity so the weights are the ones ``Upsample`` uses."""
    key = (size, str(device), dtype)
    m = _interp_cache.get(key)
    if m is None:
        eye = torch.eye(size, dtype=dtype, device=device).view(1, size, size, 1)
        m = F.interpolate(eye, size=(2 * size, 1), mode="bilinear", align_corners=False)[0, :, :, 0].t().contiguous()
        _interp_cache[key] = m
    return m


def bilinear_upsample2x(x: torch.Tensor) -> torch.Tensor:
    """``Upsample(scale_factor=2, mode="bilinear")`` as two small matrix products (separable interpolation);
    backward is two more products instead of torch's atomics scatter."""
    a_h = _interp_matrix(x.shape[2], x.device, x.dtype)
    a_w = _interp_matrix(x.shape[3], x.device, x.dtype)
    if x.is_cuda and x.dtype == torch.float64 and x.dim() == 4 and x.numel() > 0:
        from .. import circuit as _c
        return _c.upsample2x(x, a_h, a_w)        # the same two products as gathers over the non-zero weights
    return torch.einsum("Oh,bchw,Pw->bcOP", a_h, x, a_w)


def is_bilinear2x(m) -> bool:
    return isinstance(m, torch.nn.Upsample) and m.mode == "bilinear" and m.scale_factor == 2 \
        and not m.align_corners and m.size is None
