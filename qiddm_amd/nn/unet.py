"""UNet glue around the quantum convolutions (reference nn/unet.py).  Everything here is
classical PyTorch-ROCm (BatchNorm / MaxPool / bilinear Upsample / concat stay on
MIOpen / rocBLAS through torch); only ``Conv2d(qdepth > 0)`` reaches the HIP engine."""
from __future__ import annotations

import torch

from .qconv import QConv2d
from .utils import autopad, get_label_embedding


def Conv2d(**kwargs):
    """``qdepth > 0`` -> :class:`QConv2d`, else ``torch.nn.Conv2d(...).double()``
    (reference nn/unet.py:9-24; default qdepth 3)."""
    qdepth = kwargs.pop("qdepth", 3)
    if qdepth > 0:
        return QConv2d(qdepth=qdepth, **kwargs)
    return torch.nn.Conv2d(**kwargs).double()


def _bn(ch):
    return torch.nn.BatchNorm2d(ch, dtype=torch.double)


class UpBlock(torch.nn.Module):
    """Reference nn/unet.py:28-75."""

    def __init__(self, in_channels, out_channels, kernel_size=3, qdepth=3):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.up_conv = torch.nn.Sequential(
            torch.nn.Upsample(scale_factor=2, mode="bilinear"),
            Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, padding=0, qdepth=qdepth),
        ).double()
        self.net = torch.nn.Sequential(
            Conv2d(in_channels=2 * out_channels, out_channels=out_channels, kernel_size=kernel_size,
                   padding=1, qdepth=qdepth),
            torch.nn.ReLU(),
            _bn(out_channels),
            Conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=kernel_size,
                   padding=1, qdepth=qdepth),
            _bn(out_channels),
            torch.nn.ReLU(),
        ).double()

    def forward(self, from_down, from_up):
        from_up = self.up_conv(from_up)
        from_down, from_up = autopad(from_down.double(), from_up.double())
        return self.net(torch.cat([from_up, from_down], dim=1).double())


class DownBlock(torch.nn.Module):
    """Reference nn/unet.py:78-116."""

    def __init__(self, in_channels, out_channels, pooling, kernel_size=3, qdepth=3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.pooling = kernel_size, pooling
        self.net = torch.nn.Sequential(
            Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                   qdepth=qdepth, padding=1),
            _bn(out_channels),
            torch.nn.ReLU(),
            Conv2d(in_channels=out_channels, out_channels=out_channels, kernel_size=kernel_size,
                   qdepth=qdepth, padding=1),
            _bn(out_channels),
            torch.nn.ReLU(),
        ).double()
        if self.pooling:
            self.pooling_layer = torch.nn.MaxPool2d(kernel_size=2, stride=2)

    def forward(self, x):
        before_pool = self.net(x.double())
        x = self.pooling_layer(before_pool) if self.pooling else before_pool
        return x, before_pool


class UNetUndirected(torch.nn.Module):
    """Reference nn/unet.py:119-180.  ``(depth=3, start_channels=8, qdepth=3)``."""

    def __init__(self, depth=3, start_channels=8, qdepth=3):
        super().__init__()
        self.depth, self.start_channels, self.qdepth = depth, start_channels, qdepth
        assert self.depth > 0, "Depth must be greater than 0"
        ch_out = -1
        downs = []
        for i in range(depth):
            ch_in = 1 if i == 0 else ch_out
            ch_out = start_channels * 2 ** i
            downs.append(DownBlock(ch_in, ch_out, pooling=i < depth - 1, qdepth=qdepth))
        ups = []
        for _ in range(depth - 1):
            ch_in, ch_out = ch_out, ch_out // 2
            ups.append(UpBlock(ch_in, ch_out, qdepth=qdepth))
        self.down_blocks = torch.nn.ModuleList(downs).double()
        self.up_blocks = torch.nn.ModuleList(ups).double()
        self.final_conv = Conv2d(in_channels=ch_out, out_channels=1, kernel_size=1, padding=0,
                                 qdepth=qdepth).double()

    def forward(self, x):
        skips = []
        x = x.double()
        for block in self.down_blocks:
            x, before_pool = block(x)
            skips.append(before_pool)
        for i, block in enumerate(self.up_blocks):
            x = block(skips[-(i + 2)].double(), x.double())
        return self.final_conv(x)

    def extra_repr(self) -> str:
        return f"depth={self.depth}"

    def save_name(self) -> str:
        return f"unet_undirected_d{self.depth}_s{self.start_channels}_d{self.qdepth}"


class UnetDirected(UNetUndirected):
    """Label-conditioned variant (reference nn/unet.py:183-190)."""

    def forward(self, x, y):
        mask = get_label_embedding(y.double(), x.shape[2], x.shape[3])
        return super().forward(x.double() + mask)

    def save_name(self) -> str:
        return f"unet_directed_d{self.depth}_s{self.start_channels}_d{self.qdepth}"
