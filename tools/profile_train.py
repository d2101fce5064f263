"""Run the recorded training step of the flagship a few hundred times (for rocprofv3 --kernel-trace --stats)."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd import models, nn, noise
from qiddm_amd.trainer import GraphedTrainStep

mode = sys.argv[1] if len(sys.argv) > 1 else "adjoint"
torch.manual_seed(42)
net = nn.QNN_noise(784, 8, 14, detach_quantum=(mode == "detached"))
if mode == "adjoint":
    net.qnode.diff_method = "adjoint"
diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (28, 28), torch.nn.MSELoss()).to("cuda", dtype=torch.double).train()
B = int(os.environ.get("QIDDM_TRAIN_BATCH", "256"))
x = torch.rand(B, 784, dtype=torch.double, device="cuda")
from qiddm_amd.optim import FusedAdam
step = GraphedTrainStep(diff, FusedAdam(diff.parameters(), lr=1e-3), x, T=10, noise=os.environ.get("QIDDM_TRAIN_NOISE", "fused"))
for _ in range(5):
    step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print(f"{mode}: batch {B}: {dt*1e6:.1f} us/step, {B*10/dt/1e6:.2f} M img/s")
