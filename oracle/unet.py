"""Oracle restatement of ``UNetUndirectedS`` end to end (row A6 of SURVEY.md section 8a).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The quantum convolutions go through
``oracle.circuits.qconv2d_forward`` (amplitude / SEL-CNOT / probs family: **parity unpinned**, finding F3 -- the
reference's layer never calls its circuit as checked in); the wiring and the classical glue are CPU torch float64
and follow the reference lines cited below.

Functional on a ``state_dict`` with the reference's own key names (``nn/unet_simple.py:6-84`` builds the modules,
so these are the keys its checkpoints would carry):

    down_blocks.<i>.net.0.weights                      QConv2d(k=3, padding=1)           nn/unet_simple.py:9-16
    down_blocks.<i>.net.1.{weight,bias,running_*}      BatchNorm2d                       :17
    up_blocks.<i>.up_conv.1.weights                    Upsample(x2, bilinear) -> QConv2d(k=1, padding=0)   :40-49
    up_blocks.<i>.net.0.weights / .net.1.*             QConv2d(k=3, padding=1) -> BatchNorm2d              :30-39
    final_conv.{weight,bias}                           classical 1x1 Conv2d (qdepth=0 super-ctor, :59; nn/unet.py:154-160)
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import circuits as oc


def autopad(x, y):
    """``nn/utils.py:22-39``: zero-pad ``y`` to the spatial size of ``x`` (ceil on the left/top)."""
    xs, ys = x.shape, y.shape
    if tuple(xs) < tuple(ys):
        y, x = autopad(y, x)
        return x, y
    y = F.pad(y, (math.ceil((xs[3] - ys[3]) / 2), math.floor((xs[3] - ys[3]) / 2),
                  math.ceil((xs[2] - ys[2]) / 2), math.floor((xs[2] - ys[2]) / 2)), mode="constant", value=0)
    return x, y


def _bn(x, sd, prefix, training, eps=1e-5):
    """``torch.nn.BatchNorm2d`` defaults; training mode normalises with the batch statistics (biased variance)."""
    w, b = sd[prefix + ".weight"].double(), sd[prefix + ".bias"].double()
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = sd[prefix + ".running_mean"].double(), sd[prefix + ".running_var"].double()
    xn = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    return xn * w[None, :, None, None] + b[None, :, None, None]


def unet_simple_forward(x, sd, depth=3, start_channels=8, training=False):
    """``UNetUndirected.forward`` (``nn/unet.py:162-174``) over ``DownBlockS`` / ``UpBlockS``
    (``nn/unet_simple.py:6-49``; block forwards ``nn/unet.py:70-75, 111-116``)."""
    x = x.double()
    skips = []
    out_ch = -1
    for i in range(depth):
        out_ch = start_channels * 2 ** i
        x = oc.qconv2d_forward(x, sd[f"down_blocks.{i}.net.0.weights"], out_ch, (3, 3), (1, 1))
        x = _bn(x, sd, f"down_blocks.{i}.net.1", training)
        skips.append(x)
        if i < depth - 1:                                       # no pooling in the last block (nn/unet.py:142)
            x = F.max_pool2d(x, kernel_size=2, stride=2)
    for i in range(depth - 1):
        out_ch //= 2
        skip = skips[-(i + 2)]
        up = F.interpolate(x, scale_factor=2, mode="bilinear")   # torch.nn.Upsample(scale_factor=2, mode="bilinear")
        up = oc.qconv2d_forward(up, sd[f"up_blocks.{i}.up_conv.1.weights"], out_ch, (1, 1), (0, 0))
        skip, up = autopad(skip, up)
        x = torch.cat([up, skip], dim=1)
        x = oc.qconv2d_forward(x, sd[f"up_blocks.{i}.net.0.weights"], out_ch, (3, 3), (1, 1))
        x = _bn(x, sd, f"up_blocks.{i}.net.1", training)
    return F.conv2d(x, sd["final_conv.weight"].double(), sd["final_conv.bias"].double())
