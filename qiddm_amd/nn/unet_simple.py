""""Simple" UNet: every block is one QConv2d + BatchNorm2d (reference nn/unet_simple.py).

Same class names, constructor orders, attribute names (``net``, ``up_conv``, ``down_blocks``,
``up_blocks``, ``final_conv``) and ``save_name()`` strings as the reference, so its ``state_dict`` keys
line up; the layers themselves are assembled by the two helpers below."""
from __future__ import annotations

import torch

from .. import circuit as _c
from .qconv import QConv2d
from .unet import DownBlock, UNetUndirected, UpBlock
from .utils import autopad, get_label_embedding


def _qconv_then_bn(c_in: int, c_out: int, k, qdepth: int) -> torch.nn.Sequential:
    """[QConv2d(k, pad 1), BatchNorm2d] -- indices 0 and 1 of every ``net`` (reference :9-18, :30-39)."""
    conv = QConv2d(in_channels=c_in, out_channels=c_out, kernel_size=k, padding=1, qdepth=qdepth)
    return torch.nn.Sequential(conv, torch.nn.BatchNorm2d(c_out))


def _upsample_then_qconv1x1(c_in: int, c_out: int, qdepth: int) -> torch.nn.Sequential:
    """[bilinear x2, QConv2d(1x1)] -- the ``up_conv`` of an up block (reference :40-49)."""
    conv = QConv2d(in_channels=c_in, out_channels=c_out, kernel_size=1, padding=0, qdepth=qdepth)
    return torch.nn.Sequential(torch.nn.Upsample(scale_factor=2, mode="bilinear"), conv)


def _fused_conv_bn(net: torch.nn.Sequential, x):
    """Inference shortcut for a ``[QConv2d, BatchNorm2d]`` pair: one GEMM launch with the normalisation in its
    epilogue; None when not applicable (training, autograd, CPU ...)."""
    if len(net) != 2 or not isinstance(net[0], QConv2d) or not isinstance(net[1], torch.nn.BatchNorm2d):
        return None
    return net[0].eval_forward(x, batch_norm=net[1])


class DownBlockS(DownBlock):
    """Reference nn/unet_simple.py:6-18 (the classical base is built first, then ``net`` is replaced)."""

    def __init__(self, in_channels, out_channels, pooling, kernel_size=3, qdepth=3):
        super().__init__(in_channels, out_channels, pooling, kernel_size, qdepth)
        self.net = _qconv_then_bn(in_channels, out_channels, kernel_size, qdepth)

    def forward(self, x):
        before_pool = _fused_conv_bn(self.net, x)
        if before_pool is None:
            return super().forward(x)
        return (_c.max_pool2(self.pooling_layer, before_pool) if self.pooling else before_pool), before_pool


class UpBlockS(UpBlock):
    """Reference nn/unet_simple.py:21-49 (base built with qdepth=0, both sub-nets replaced)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, qdepth=3):
        super().__init__(in_channels, out_channels, kernel_size, qdepth=0)
        self.net = _qconv_then_bn(2 * out_channels, out_channels, kernel_size, qdepth)
        self.up_conv = _upsample_then_qconv1x1(in_channels, out_channels, qdepth)

    def forward(self, from_down, from_up):
        up = None
        if len(self.up_conv) == 2 and isinstance(self.up_conv[0], torch.nn.Upsample) \
                and isinstance(self.up_conv[1], QConv2d) and self.up_conv[0].mode == "bilinear" \
                and self.up_conv[0].scale_factor == 2 and not self.up_conv[0].align_corners:
            up = self.up_conv[1].eval_forward(from_up, upsample2x=True)      # the x2 is read on the fly
        if up is None:
            return super().forward(from_down, from_up)
        skip, up = autopad(from_down.to(torch.double), up)
        cat = torch.cat([up, skip], dim=1)
        out = _fused_conv_bn(self.net, cat)
        return self.net(cat) if out is None else out


class UNetUndirectedS(UNetUndirected):
    """Reference nn/unet_simple.py:52-84; ``final_conv`` stays the classical 1x1 conv of the qdepth=0 base."""

    def __init__(self, depth=3, start_channels=8, qdepth=3):
        super().__init__(depth, start_channels, qdepth=0)
        self.qdepth = qdepth
        downs = [DownBlockS(b.in_channels, b.out_channels, b.pooling, b.kernel_size, qdepth) for b in self.down_blocks]
        ups = [UpBlockS(b.in_channels, b.out_channels, b.kernel_size, qdepth) for b in self.up_blocks]
        self.down_blocks = torch.nn.ModuleList(downs)
        self.up_blocks = torch.nn.ModuleList(ups)

    def forward(self, x):
        fc = self.final_conv
        if torch.is_grad_enabled() or not x.is_cuda or not isinstance(fc, torch.nn.Conv2d) \
                or fc.kernel_size != (1, 1) or fc.stride != (1, 1) or fc.padding != (0, 0) or fc.groups != 1:
            return super().forward(x)
        # inference: same wiring, the classical 1x1 head through the float64 HIP kernel
        x = x.to(torch.double)
        skips = []
        for block in self.down_blocks:
            x, before_pool = block(x)
            skips.append(before_pool)
        for i, block in enumerate(self.up_blocks):
            x = block(skips[-(i + 2)], x)
        return _c.conv1x1_forward(x, fc.weight, fc.bias)

    def save_name(self) -> str:
        return f"unet_s_undirected_d{self.depth}_s{self.start_channels}_d{self.qdepth}"


class UnetDirectedS(UNetUndirectedS):
    """Label-conditioned variant (reference nn/unet_simple.py:87-94)."""

    def forward(self, x, y):
        return super().forward(x + get_label_embedding(y, x.shape[2], x.shape[3]))

    def save_name(self) -> str:
        return f"unet_s_directed_d{self.depth}_s{self.start_channels}_d{self.qdepth}"
