"""Oracle restatement of the denoise loop around the quantum layers.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  **Pinned**: checked
against outputs of the reference's own ``src/noise.py`` / ``src/models.py``
(fixtures ``tests/golden/diffusion_*.npz``, generator
``tests/golden/make_diffusion_golden.py``).
"""
from __future__ import annotations

import torch


def noise_weighting(tau: int, decay_mod: float, dtype=torch.float32) -> torch.Tensor:
    """``linspace(0,1,tau)**decay_mod / max`` (``src/noise.py:118-119``)."""
    w = torch.linspace(0, 1, tau, dtype=dtype) ** decay_mod
    return w / w.max()


def add_normal_noise_multiple(data, tau: int, decay_mod: float = 1.0, noise=None):
    """``src/noise.py:105-126``.  ``noise`` (B, P) may be injected; otherwise it is
    drawn exactly as the reference does (float32, default CPU generator,
    N(0.5, 0.2), ``:113-115``).  Returns ``((batch tau), pixels)``."""
    if data.dim() == 1:
        data = data.unsqueeze(0)
    batch, pixels = data.shape
    if noise is None:
        noise = torch.normal(mean=0.5, std=0.2, size=(batch, pixels))
    noise = noise.to(data.device)
    w = noise_weighting(tau, decay_mod).to(data.device).reshape(tau, 1, 1)
    noisy = data.unsqueeze(0) * (1 - w) + noise.unsqueeze(0) * w      # (tau, B, P)
    noisy = noisy.clamp(0, 1)
    return noisy.permute(1, 0, 2).reshape(batch * tau, pixels)


def training_pairs(x, T: int, shape, noise=None):
    """Noisy/clean batches of ``Diffusion.run_training_step_*`` (``src/models.py:45-63``)."""
    wd, ht = shape
    whole = add_normal_noise_multiple(x, tau=T + 1, decay_mod=3.0, noise=noise)
    whole = whole.reshape(x.shape[0], T + 1, -1)
    noisy = whole[:, 1:, :].reshape(-1, 1, wd, ht)
    clean = whole[:, :-1, :].reshape(-1, 1, wd, ht)
    return noisy, clean


def training_loss(net, x, T: int, shape, goal: str = "data", noise=None):
    """Mean-MSE loss of one training step (``src/models.py:64-66`` "data" goal,
    ``:94-98`` "noise" goal).  The caller runs ``.backward()``."""
    noisy, clean = training_pairs(x, T, shape, noise)
    pred = net(noisy)
    if goal == "data":
        return ((pred - clean) ** 2).mean(), pred
    pred = (pred - 0.5) * 0.1
    return ((pred - (noisy - clean)) ** 2).mean(), torch.clamp(noisy - pred, 0, 1)


def denoise_step(net, x, goal: str = "data", noise_factor: float = 1.0):
    """One body of the ``Diffusion.sample`` loop (``src/models.py:127-134``)."""
    predicted = net(x)
    if goal == "data":
        return predicted
    return torch.clamp(x - (predicted - 0.5) * 0.1 * noise_factor, 0, 1)


def sample(net, first_x, n_iters: int, goal: str = "data", only_last=False, step=1,
           noise_factor: float = 1.0):
    """``Diffusion.sample`` (``src/models.py:106-147``): mosaic
    ``(iters height) (batch width)``."""
    outp = [first_x]
    x = first_x
    with torch.no_grad():
        for i in range(n_iters):
            x = denoise_step(net, x, goal, noise_factor)
            if i % step == 0:
                outp.append(x)
    if only_last:
        return outp[-1]
    st = torch.stack(outp)                       # iters batch 1 height width
    it, b, _, h, w = st.shape
    return st[:, :, 0].permute(0, 2, 1, 3).reshape(it * h, b * w)
