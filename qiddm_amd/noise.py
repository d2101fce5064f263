"""Forward-noising schedule of the denoise loop (reference src/noise.py:105-126; the other
three schedules in that file are never referenced by any driver)."""
from __future__ import annotations

import torch


def add_normal_noise_multiple(data, tau, decay_mod=1.0):
    """Blend every sample with ONE draw of N(0.5, 0.2) noise at ``tau`` strengths
    ``linspace(0,1,tau)**decay_mod`` (normalised), clamp to [0,1], return
    ``((batch tau), pixels)`` batch-major.

    RNG parity with the reference: the noise is drawn float32 on the default CPU
    generator and then moved to ``data.device`` (src/noise.py:113-115)."""
    if data.dim() == 1:
        data = data.unsqueeze(0)
    batch, pixels = data.shape
    noise = torch.normal(mean=0.5, std=0.2, size=(batch, pixels)).to(data.device)
    w = torch.linspace(0, 1, tau).to(data.device) ** decay_mod
    w = (w / w.max()).reshape(1, tau, 1)
    noisy = data.unsqueeze(1) * (1 - w) + noise.unsqueeze(1) * w        # (batch, tau, pixels)
    return noisy.clamp(0, 1).reshape(batch * tau, pixels)
