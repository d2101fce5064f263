"""Denoise-step rate of the reference's shipped dense configurations (src/mnist_exm.py:45-49, src/fashion_exm.py:45) at
batch 256, f32 and f64, through bench.py's own Runner (15 steps per launch, hipGraph replay).
A/B of the lean sampler:  QIDDM_NO_LEAN_SAMPLER=1 python tools/bench_small_nets.py ;  the "noise" goal:  QIDDM_GOAL=noise ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import qiddm_amd  # noqa: E402
from qiddm_amd import models, nn, noise  # noqa: E402

GOAL = os.environ.get("QIDDM_GOAL", "data")
dev = torch.device("cuda:0")
x = (torch.rand(256, 1, 28, 28, dtype=torch.double) * 0.75 + 0.5).to(dev)
print("goal:", GOAL, " lean sampler:", "off" if os.environ.get("QIDDM_NO_LEAN_SAMPLER") == "1" else "on")
for prec in ("f32", "f64"):
    qiddm_amd.set_default_precision(prec)
    for name, ctor in (("QNN_noise(784,8,14)", lambda: nn.QNN_noise(784, 8, 14)),
                       ("QIDDM_LL_noise(784,8,6,2)", lambda: nn.QIDDM_LL_noise(784, 8, 6, 2)),
                       ("QIDDM_LL_noise(784,6,14,2)", lambda: nn.QIDDM_LL_noise(784, 6, 14, 2)),
                       ("QNN_noise(784,6,14)", lambda: nn.QNN_noise(784, 6, 14)),
                       ("QNN_noise(784,10,6)", lambda: nn.QNN_noise(784, 10, 6))):
        torch.manual_seed(42)
        net = ctor().to(dev, dtype=torch.double).eval()
        diff = models.Diffusion(net, noise.add_normal_noise_multiple, GOAL, (28, 28)).to(dev, dtype=torch.double).eval()
        try:
            r = bench.Runner(diff, x, True, 15)
            r.prepare(75)
            r.run(150)
            t = bench._time_fn(lambda: r.run(75), 10) / 75
            with torch.no_grad():
                us = bench._graph_event_us(lambda: diff.denoise_steps(x, 20)) / 20
            print(f"{prec} {name:32s} {t*1e6:8.2f} us/step wall  {us:8.2f} us/step kernel (20-step launch)  "
                  f"{256/t/1e6:7.2f} M images/s", flush=True)
        except Exception as e:
            print(f"{prec} {name:32s} failed: {e!r}")
qiddm_amd.set_default_precision("f32")
