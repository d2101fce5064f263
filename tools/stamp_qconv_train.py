#!/usr/bin/env python
"""Diagnostic: per-phase s_memtime sums (workgroup 0, thread 0) of the MFMA thin-product backward for the unet_simple
layer shapes at 256 x tau 10 samples, next to the launch duration."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
from qiddm_amd import _capi  # noqa: E402
_capi.check(_capi.lib().qiddm_set_stamp_buffer(buf.data_ptr(), buf.numel()))
from qiddm_amd import nn  # noqa: E402

torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
for (cin, cout, k, pad, side) in [(1, 8, 3, 1, 28), (16, 8, 1, 0, 28), (16, 8, 3, 1, 28), (8, 16, 3, 1, 14),
                                  (32, 16, 1, 0, 14), (32, 16, 3, 1, 14), (16, 32, 3, 1, 7)]:
    layer = nn.QConv2d(cin, cout, kernel_size=k, padding=pad, qdepth=3).to("cuda").train()
    x = torch.rand(B, cin, side, side, dtype=torch.float64, device="cuda", requires_grad=True)
    y = layer(x)
    g = torch.randn_like(y)
    for _ in range(2):
        torch.autograd.grad(y, [x, layer.weights], g, retain_graph=True)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    t0 = time.perf_counter()
    for _ in range(5):
        torch.autograd.grad(y, [x, layer.weights], g, retain_graph=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    n = max(t[5], 1)
    print(f"C{cin}->{cout} k{k} {side}x{side} (F={cin*k*k}): tiles/WG {t[5]}  per tile: gather {t[0]//n}  prod1+epi {t[1]//n}  "
          f"prod2+store {t[2]//n}  next-gather issue {t[4]//n}  prod3 {t[3]//n} ticks;  whole backward {dt*1e3:.3f} ms")
