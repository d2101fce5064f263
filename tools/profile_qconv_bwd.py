"""Run the unitary-route backward of one unet_simple convolution a few times (for rocprofv3 --kernel-trace / --pmc).
usage: profile_qconv_bwd.py [cin cout k pad side [batch]]   default: 16 8 3 1 28 2560"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd import nn  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
cin, cout, k, pad, side = (a + [16, 8, 3, 1, 28][len(a):])[:5]
B = a[5] if len(a) > 5 else 2560
torch.manual_seed(0)
layer = nn.QConv2d(cin, cout, kernel_size=k, padding=pad, qdepth=3).to("cuda").train()
x = torch.rand(B, cin, side, side, dtype=torch.float64, device="cuda", requires_grad=True)
y = layer(x)
g = torch.randn_like(y)
for _ in range(2):
    torch.autograd.grad(y, [x, layer.weights], g, retain_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    torch.autograd.grad(y, [x, layer.weights], g, retain_graph=True)
torch.cuda.synchronize()
print(f"C{cin}->{cout} k{k} {side}x{side} B={B}: backward {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
