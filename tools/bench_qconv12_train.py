"""Training step (forward + backward) of BASELINE config 4's layer -- 12-qubit QConv2d(256 -> 256, 3x3, qdepth 3) on
(B, 256, 32, 32) -- through the circuit unitary with library GEMMs, next to the per-pixel wide adjoint."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
from qiddm_amd import nn, set_default_precision
from qiddm_amd import circuit as qc

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(42)
layer = nn.QConv2d(256, 256, qdepth=3).to("cuda").train()
x = torch.rand(batch, 256, 32, 32, dtype=torch.double, device="cuda").requires_grad_(True)
g = torch.randn(batch, 256, 32, 32, dtype=torch.double, device="cuda")


def step():
    layer.weights.grad = None
    x.grad = None
    y = layer(x)
    (y * g).sum().backward()


def timeit(f, n):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t = timeit(step, 3)
px = batch * 1024
flop = 4 * 2.0 * 2304 * 512 * px          # forward + three backward products
print(f"unitary route: {t*1e3:.1f} ms per fwd+bwd of {px} pixels ({px/t/1e6:.2f} M pixels/s, {flop/t/1e12:.1f} TFLOP/s "
      f"over the four products)")
gw = layer.weights.grad.clone()
# per-pixel route on one image (float64 setting -> tiled forward + wide adjoint), extrapolated
set_default_precision("f64")
x1 = x[:1].detach().requires_grad_(True)


def step1():
    layer.weights.grad = None
    y = layer(x1)
    (y * g[:1]).sum().backward()


t1 = timeit(step1, 1)
set_default_precision("f32")
print(f"per-pixel sweep (float64): {t1*1e3:.1f} ms per image -> {t1*batch*1e3:.0f} ms for the batch ({t1*batch/t:.0f}x)")
