"""Denoise-step rate of the reference's other shipped dense configurations (src/mnist_exm.py:45-49) at batch 256."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from qiddm_amd import models, nn, noise

dev = torch.device("cuda:0")
x = (torch.rand(256, 1, 28, 28, dtype=torch.double) * 0.75 + 0.5).to(dev)
for name, ctor in (("QNN_noise(784,8,14)", lambda: nn.QNN_noise(784, 8, 14)),
                   ("QIDDM_LL_noise(784,6,14,2)", lambda: nn.QIDDM_LL_noise(784, 6, 14, 2)),
                   ("QIDDM_LL_noise(784,8,6,2)", lambda: nn.QIDDM_LL_noise(784, 8, 6, 2)),
                   ("QNN_noise(784,6,14)", lambda: nn.QNN_noise(784, 6, 14)),
                   ("QNN_noise(784,10,6)", lambda: nn.QNN_noise(784, 10, 6))):
    torch.manual_seed(42)
    net = ctor().to(dev, dtype=torch.double).eval()
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (28, 28)).to(dev, dtype=torch.double).eval()
    try:
        run, _ = bench.make_runner(diff, x, True, 15)
        run(150)
        t = bench._time_fn(lambda: run(75), 10) / 75
        print(f"{name:32s} {t*1e6:8.2f} us/step  {256/t/1e6:7.2f} M images/s")
    except Exception as e:
        print(f"{name:32s} failed: {e!r}")
