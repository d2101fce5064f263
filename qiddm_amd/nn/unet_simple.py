""""Simple" UNet: every block is one QConv2d + BatchNorm2d (reference nn/unet_simple.py)."""
from __future__ import annotations

import torch

from .qconv import QConv2d
from .unet import DownBlock, UNetUndirected, UpBlock
from .utils import get_label_embedding


class DownBlockS(DownBlock):
    """Reference nn/unet_simple.py:6-18."""

    def __init__(self, in_channels, out_channels, pooling, kernel_size=3, qdepth=3):
        super().__init__(in_channels, out_channels, pooling, kernel_size, qdepth)
        self.net = torch.nn.Sequential(
            QConv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                    qdepth=qdepth, padding=1),
            torch.nn.BatchNorm2d(out_channels),
        )


class UpBlockS(UpBlock):
    """Reference nn/unet_simple.py:21-49 (the base is built classically, qdepth=0, then
    both sub-nets are replaced)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, qdepth=3):
        super().__init__(in_channels, out_channels, kernel_size, qdepth=0)
        self.net = torch.nn.Sequential(
            QConv2d(in_channels=2 * out_channels, out_channels=out_channels, kernel_size=kernel_size,
                    padding=1, qdepth=qdepth),
            torch.nn.BatchNorm2d(out_channels),
        )
        self.up_conv = torch.nn.Sequential(
            torch.nn.Upsample(scale_factor=2, mode="bilinear"),
            QConv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=1, padding=0,
                    qdepth=qdepth),
        )


class UNetUndirectedS(UNetUndirected):
    """Reference nn/unet_simple.py:52-84; ``final_conv`` stays the classical 1x1 conv of the
    qdepth=0 base."""

    def __init__(self, depth=3, start_channels=8, qdepth=3):
        super().__init__(depth, start_channels, qdepth=0)
        self.qdepth = qdepth
        self.down_blocks = torch.nn.ModuleList(
            DownBlockS(db.in_channels, db.out_channels, db.pooling, db.kernel_size, qdepth)
            for db in self.down_blocks)
        self.up_blocks = torch.nn.ModuleList(
            UpBlockS(ub.in_channels, ub.out_channels, ub.kernel_size, qdepth) for ub in self.up_blocks)

    def save_name(self) -> str:
        return f"unet_s_undirected_d{self.depth}_s{self.start_channels}_d{self.qdepth}"


class UnetDirectedS(UNetUndirectedS):
    """Reference nn/unet_simple.py:87-94."""

    def forward(self, x, y):
        mask = get_label_embedding(y, x.shape[2], x.shape[3])
        return super().forward(x + mask)

    def save_name(self) -> str:
        return f"unet_s_directed_d{self.depth}_s{self.start_channels}_d{self.qdepth}"
