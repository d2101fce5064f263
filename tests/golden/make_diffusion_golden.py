"""Generate tests/golden/diffusion_*.npz by RUNNING the reference's own
``src/noise.py`` and ``src/models.py`` (torch + einops + tqdm only; importable
in the build container).  Run once from the repo root:

    python tests/golden/make_diffusion_golden.py

Only the resulting small tensors are committed; the reference's source never
enters this repository and ``/root/reference`` is never read by the tests.
The stub net below is ours: a two-parameter pixelwise affine+sigmoid map, so
that the loss, the reconstruction AND the gradients that ``Diffusion`` leaves
in ``.grad`` (it calls ``.backward()`` internally, ``src/models.py:67,99``) are
all pinned.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
sys.path.insert(0, REF)
import models as ref_models  # noqa: E402
import noise as ref_noise  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


class StubNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.tensor(1.7, dtype=torch.float64))
        self.b = torch.nn.Parameter(torch.tensor(-0.6, dtype=torch.float64))

    def forward(self, x):
        return torch.sigmoid(self.a * x + self.b)

    def save_name(self):
        return "stub"


def main():
    # --- A8: add_normal_noise_multiple --------------------------------------
    torch.manual_seed(1234)
    x = torch.rand(5, 64, dtype=torch.float64)
    torch.manual_seed(99)
    noise_draw = torch.normal(mean=0.5, std=0.2, size=(5, 64))      # same call => same stream
    torch.manual_seed(99)
    noisy = ref_noise.add_normal_noise_multiple(x, tau=11, decay_mod=3.0)
    torch.manual_seed(99)
    noisy_d1 = ref_noise.add_normal_noise_multiple(x, tau=4, decay_mod=1.0)
    np.savez_compressed(
        os.path.join(OUT, "diffusion_noise.npz"),
        x=x.numpy(), noise=noise_draw.numpy(), noisy_tau11_decay3=noisy.numpy(),
        noisy_tau4_decay1=noisy_d1.numpy(), seed_x=1234, seed_noise=99,
    )

    # --- A7: Diffusion training step, both goals ----------------------------
    out = {}
    for goal in ("data", "noise"):
        net = StubNet()
        diff = ref_models.Diffusion(net=net, noise_f=ref_noise.add_normal_noise_multiple,
                                    prediction_goal=goal, shape=(8, 8),
                                    loss=torch.nn.MSELoss()).to(dtype=torch.double)
        diff.train()
        torch.manual_seed(7)
        xb = torch.rand(4, 64, dtype=torch.float64)
        torch.manual_seed(21)
        nz = torch.normal(mean=0.5, std=0.2, size=(4, 64))
        torch.manual_seed(21)
        res = diff(x=xb, T=10, verbose=True)
        out[f"{goal}_x"] = xb.numpy()
        out[f"{goal}_noise"] = nz.numpy()
        out[f"{goal}_loss"] = res[0].detach().numpy()
        out[f"{goal}_recon"] = res[1].detach().numpy()
        out[f"{goal}_grad_a"] = net.a.grad.numpy()
        out[f"{goal}_grad_b"] = net.b.grad.numpy()
        out[f"{goal}_save_name"] = diff.save_name()
        # --- sampling loop -----------------------------------------------
        diff.eval()
        torch.manual_seed(3)
        first_x = torch.rand(10, 1, 8, 8, dtype=torch.float64) * 0.75 + 0.5
        mosaic = diff.sample(first_x=first_x, n_iters=5, show_progress=False, only_last=False)
        last = diff.sample(first_x=first_x, n_iters=5, show_progress=False, only_last=True)
        out[f"{goal}_first_x"] = first_x.numpy()
        out[f"{goal}_mosaic"] = mosaic.numpy()
        out[f"{goal}_last"] = last.numpy()
    np.savez_compressed(os.path.join(OUT, "diffusion_step.npz"), **out)
    print("wrote", os.listdir(OUT))


if __name__ == "__main__":
    main()
