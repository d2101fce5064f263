"""A4 net of BASELINE config 3 (QDenseUndirected_old_noise(60, 28), src/mnist_exm.py:45) at inference: the unitary route
(one float32 product with the cached circuit unitary) against the per-sample simulation kernel, HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qiddm_amd import nn  # noqa: E402
from qiddm_amd.nn import qdense  # noqa: E402

dev = "cuda"
torch.manual_seed(42)
net = nn.QDenseUndirected_old_noise(60, 28).to(dev).eval()
for batch in (256, 1024, 4096):
    x = torch.rand(batch, 1, 28, 28, dtype=torch.double, device=dev)
    with torch.no_grad():
        us = bench._graph_event_us(lambda: net(x), launches=10)
        qdense._DENSE_UNITARY = False
        us0 = bench._graph_event_us(lambda: net(x), launches=10)
        qdense._DENSE_UNITARY = True
    print(f"QDenseUndirected_old_noise(60,28) batch {batch}: unitary route {us:.1f} us ({batch/us:.2f} M img/s), "
          f"simulation {us0:.1f} us ({batch/us0:.2f} M img/s)", flush=True)
