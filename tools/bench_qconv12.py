#!/usr/bin/env python
"""BASELINE config 4's layer (12-qubit QConv2d, C_in = C_out = 256, 3x3, qdepth 3) on a (B, 256, 32, 32) batch:
eval-mode route (circuit unitary once + MFMA GEMM) against the tiled circuit simulation (one workgroup per pixel)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd import nn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
layer = nn.QConv2d(256, 256, qdepth=3).to("cuda")
x = torch.rand(B, 256, 32, 32, dtype=torch.float64, device="cuda")


def timeit(n):
    with torch.no_grad():
        layer(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            layer(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


layer.eval()
t0 = time.perf_counter()
with torch.no_grad():
    layer(x)
torch.cuda.synchronize()
t_first = time.perf_counter() - t0
t_gemm = timeit(10)
layer.train()
t_sim = timeit(3)
px = B * 32 * 32
flop = 2.0 * 2304 * 512 * px
print(f"12-qubit QConv2d(256,256,3x3,qdepth 3), {px} output pixels: unitary+GEMM {t_gemm*1e3:.3f} ms "
      f"({px/t_gemm/1e6:.1f} M px/s, {flop/t_gemm/1e12:.1f} TFLOP/s f32 MFMA; first call incl. the 4096x4096 unitary "
      f"{t_first*1e3:.1f} ms); tiled simulation {t_sim*1e3:.3f} ms ({px/t_sim/1e6:.1f} M px/s)")
