"""UNet glue around the quantum convolutions (reference nn/unet.py).  Everything here is classical
PyTorch-ROCm (BatchNorm / MaxPool / bilinear Upsample / concat stay on MIOpen / rocBLAS through torch);
only ``Conv2d(qdepth > 0)`` reaches the HIP engine.  Module attribute names and Sequential positions
match the reference so that its checkpoints (``net.down_blocks.0.net.0.weight`` ...) load unchanged."""
from __future__ import annotations

import torch

from .. import circuit as _c
from .qconv import QConv2d
from .utils import (autopad, bilinear_upsample2x, get_label_embedding, is_bilinear2x, is_pointwise,
                    pointwise_conv)

F64 = torch.double


def Conv2d(**kwargs):
    """``qdepth > 0`` -> :class:`QConv2d`, else ``torch.nn.Conv2d(...).double()``
    (reference nn/unet.py:9-24; default qdepth 3)."""
    qdepth = kwargs.pop("qdepth", 3)
    return QConv2d(qdepth=qdepth, **kwargs) if qdepth > 0 else torch.nn.Conv2d(**kwargs).to(F64)


def _conv3(c_in, c_out, k, qdepth):
    return Conv2d(in_channels=c_in, out_channels=c_out, kernel_size=k, padding=1, qdepth=qdepth)


def _run(net: torch.nn.Sequential, x):
    """``net(x)`` with training-mode float64 BatchNorm2d layers on the HIP kernels (same numbers, same running
    statistics; torch's float64 batch norm is several generic launches per direction)."""
    i = 0
    while i < len(net):
        m = net[i]
        if isinstance(m, QConv2d) and i + 1 < len(net) and type(net[i + 1]) is torch.nn.BatchNorm2d:
            # [QConv2d, BatchNorm2d] (unet_simple): one autograd node, the BatchNorm backward inside the convolution's
            fused = m.train_forward_bn(x, net[i + 1])
            if fused is not None:
                x, i = fused, i + 2
                continue
        x = _c.batch_norm_train(m, x) if type(m) is torch.nn.BatchNorm2d else m(x)
        i += 1
    return x


class UpBlock(torch.nn.Module):
    """Reference nn/unet.py:28-75.  ``up_conv`` = [Upsample, 1x1 conv];
    ``net`` = [conv, ReLU, BN, conv, BN, ReLU] (that order, as in the reference)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, qdepth=3):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        one_by_one = Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, padding=0, qdepth=qdepth)
        self.up_conv = torch.nn.Sequential(torch.nn.Upsample(scale_factor=2, mode="bilinear"), one_by_one).to(F64)
        self.net = torch.nn.Sequential(
            _conv3(2 * out_channels, out_channels, kernel_size, qdepth), torch.nn.ReLU(),
            torch.nn.BatchNorm2d(out_channels, dtype=F64),
            _conv3(out_channels, out_channels, kernel_size, qdepth),
            torch.nn.BatchNorm2d(out_channels, dtype=F64), torch.nn.ReLU(),
        ).to(F64)

    def _up(self, x):
        if len(self.up_conv) == 2 and is_bilinear2x(self.up_conv[0]) and x.is_cuda:
            x = bilinear_upsample2x(x.to(F64))          # same interpolation weights, as two matrix products
            conv = self.up_conv[1]
            return pointwise_conv(conv, x) if is_pointwise(conv) else conv(x)
        return self.up_conv(x)

    def forward(self, from_down, from_up):
        skip, up = autopad(from_down.to(F64), self._up(from_up).to(F64))
        return _run(self.net, torch.cat([up, skip], dim=1).to(F64))


class DownBlock(torch.nn.Module):
    """Reference nn/unet.py:78-116.  ``net`` = [conv, BN, ReLU, conv, BN, ReLU]; returns
    (pooled, before_pool)."""

    def __init__(self, in_channels, out_channels, pooling, kernel_size=3, qdepth=3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.pooling = kernel_size, pooling
        self.net = torch.nn.Sequential(
            _conv3(in_channels, out_channels, kernel_size, qdepth),
            torch.nn.BatchNorm2d(out_channels, dtype=F64), torch.nn.ReLU(),
            _conv3(out_channels, out_channels, kernel_size, qdepth),
            torch.nn.BatchNorm2d(out_channels, dtype=F64), torch.nn.ReLU(),
        ).to(F64)
        if pooling:
            self.pooling_layer = torch.nn.MaxPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        before_pool = _run(self.net, x.to(F64))
        return (_c.max_pool2(self.pooling_layer, before_pool) if self.pooling else before_pool), before_pool


class UNetUndirected(torch.nn.Module):
    """Reference nn/unet.py:119-180.  ``(depth=3, start_channels=8, qdepth=3)``: channels
    start, 2*start, ... on the way down, halved on the way up, 1x1 ``final_conv`` to one channel."""

    def __init__(self, depth=3, start_channels=8, qdepth=3):
        super().__init__()
        self.depth, self.start_channels, self.qdepth = depth, start_channels, qdepth
        assert self.depth > 0, "Depth must be greater than 0"
        widths = [start_channels * 2 ** i for i in range(depth)]
        downs = [DownBlock(1 if i == 0 else widths[i - 1], widths[i], pooling=i < depth - 1, qdepth=qdepth)
                 for i in range(depth)]
        ups = [UpBlock(widths[i], widths[i] // 2, qdepth=qdepth) for i in range(depth - 1, 0, -1)]
        self.down_blocks = torch.nn.ModuleList(downs).to(F64)
        self.up_blocks = torch.nn.ModuleList(ups).to(F64)
        last = widths[0] if depth > 1 else widths[-1]
        self.final_conv = Conv2d(in_channels=last, out_channels=1, kernel_size=1, padding=0, qdepth=qdepth).to(F64)

    def forward(self, x):
        x = x.to(F64)
        skips = []
        for block in self.down_blocks:
            x, before_pool = block(x)
            skips.append(before_pool)
        for i, block in enumerate(self.up_blocks):
            x = block(skips[-(i + 2)].to(F64), x.to(F64))
        if is_pointwise(self.final_conv) and x.is_cuda:
            return pointwise_conv(self.final_conv, x)
        return self.final_conv(x)

    def extra_repr(self) -> str:
        return f"depth={self.depth}"

    def save_name(self) -> str:
        return f"unet_undirected_d{self.depth}_s{self.start_channels}_d{self.qdepth}"


class UnetDirected(UNetUndirected):
    """Label-conditioned variant (reference nn/unet.py:183-190)."""

    def forward(self, x, y):
        return super().forward(x.to(F64) + get_label_embedding(y.to(F64), x.shape[2], x.shape[3]))

    def save_name(self) -> str:
        return f"unet_directed_d{self.depth}_s{self.start_channels}_d{self.qdepth}"
