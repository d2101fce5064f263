#!/usr/bin/env python
"""Diagnostic: phase stamps (s_memtime of block 0 / thread 0) of the quantum-convolution GEMM kernel and its
launch duration for the unet_simple layer shapes at batch 256."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
from qiddm_amd import _capi  # noqa: E402
_capi.check(_capi.lib().qiddm_set_stamp_buffer(buf.data_ptr(), buf.numel()))
from qiddm_amd import nn  # noqa: E402

torch.manual_seed(0)
for (cin, cout, k, pad, side, up) in [(1, 8, 3, 1, 28, False), (16, 8, 1, 0, 14, True), (16, 8, 3, 1, 28, False),
                                      (8, 16, 3, 1, 14, False), (32, 16, 3, 1, 14, False), (16, 32, 3, 1, 7, False)]:
    layer = nn.QConv2d(cin, cout, kernel_size=k, padding=pad, qdepth=3).to("cuda").eval()
    x = torch.rand(256, cin, side, side, dtype=torch.float64, device="cuda")
    with torch.no_grad():
        for _ in range(3):
            layer.eval_forward(x, upsample2x=up)
        torch.cuda.synchronize()
        t = buf.cpu().tolist()
        t0 = time.perf_counter()
        for _ in range(50):
            layer.eval_forward(x, upsample2x=up)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
    d = [t[i + 1] - t[i] for i in range(4)]
    print(f"C{cin}->{cout} k{k} {side}x{side}{' x2' if up else ''}: setup {d[0]}  K-loop {d[1]}  epilogue {d[2]}  store {d[3]} "
          f"ticks;  pack+gemm {dt*1e6:.1f} us per call")
