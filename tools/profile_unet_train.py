"""unet_simple training step (Diffusion loss + backward + Adam) with a torch.profiler kernel table."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd import nn
from qiddm_amd.models import Diffusion

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
with_table = "--no-table" not in sys.argv       # under rocprofv3, skip torch's own profiler
tau = 10
torch.manual_seed(42)
net = nn.UNetUndirectedS(3, 8, 3).to("cuda").to(torch.double).train()
from qiddm_amd.noise import add_normal_noise_multiple
diff = Diffusion(net=net, noise_f=add_normal_noise_multiple, prediction_goal="data", shape=(28, 28))
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
x = torch.rand(batch, 784, dtype=torch.double, device="cuda")


def step():
    opt.zero_grad()
    loss = diff(x=x, T=tau, verbose=False)
    opt.step()
    return loss


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for _ in range(n):
    step()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / n
print(f"UNetUndirectedS(3,8,3) training step, batch {batch} x tau {tau}: {t*1e3:.1f} ms ({batch*tau/t:.0f} images/s)")
if with_table:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))

# the same step recorded into HIP graphs (device noise stream, one-launch Adam)
from qiddm_amd.optim import FusedAdam
from qiddm_amd.trainer import GraphedTrainStep
torch.manual_seed(42)
net2 = nn.UNetUndirectedS(3, 8, 3).to("cuda").to(torch.double).train()
diff2 = Diffusion(net=net2, noise_f=add_normal_noise_multiple, prediction_goal="data", shape=(28, 28))
gstep = GraphedTrainStep(diff2, FusedAdam(net2.parameters(), lr=1e-3), x, T=tau, noise="device")
gstep(x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    loss = gstep(x)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 10
print(f"recorded in HIP graphs: {t*1e3:.1f} ms ({batch*tau/t:.0f} images/s), loss {float(loss[0]):.6f}")
