"""unet_simple inference, eval-mode (unitary + MFMA GEMM convolutions) vs circuit-simulation convolutions."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd import nn

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(42)
net = nn.UNetUndirectedS(3, 8, 3).to("cuda").to(torch.double).eval()
x = torch.rand(batch, 1, 28, 28, dtype=torch.double, device="cuda") * 0.75 + 0.5


def timeit(f, n):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    t_gemm = timeit(lambda: net(x), 20)
    for m in net.modules():
        if isinstance(m, nn.QConv2d):
            m.training = True
    t_sim = timeit(lambda: net(x), 5)
print(f"UNetUndirectedS(3,8,3) batch {batch}: unitary+GEMM {t_gemm*1e3:.3f} ms ({batch/t_gemm:.0f} img/s), "
      f"circuit simulation {t_sim*1e3:.3f} ms ({batch/t_sim:.0f} img/s)")
