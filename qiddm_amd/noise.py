"""Forward-noising schedule of the denoise loop (reference src/noise.py:105-126; the other
three schedules in that file are never referenced by any driver)."""
from __future__ import annotations

import torch


_SCHEDULES = {}


def _schedule(tau, decay_mod, device):
    """``linspace(0,1,tau).to(device)**decay_mod / max`` as (1, tau, 1); computed once per key, in the
    reference's order of operations, so replays inside a recorded HIP graph need no host transfer."""
    key = (int(tau), float(decay_mod), str(device))
    w = _SCHEDULES.get(key)
    if w is None:
        w = torch.linspace(0, 1, tau).to(device) ** decay_mod
        w = _SCHEDULES[key] = (w / w.max()).reshape(1, tau, 1)
    return w


def add_normal_noise_multiple(data, tau, decay_mod=1.0, noise=None):
    """Blend every sample with ONE draw of N(0.5, 0.2) noise at ``tau`` strengths
    ``linspace(0,1,tau)**decay_mod`` (normalised), clamp to [0,1], return
    ``((batch tau), pixels)`` batch-major.

    RNG parity with the reference: the noise is drawn float32 on the default CPU
    generator and then moved to ``data.device`` (src/noise.py:113-115).  ``noise`` (an extension) hands in a
    pre-drawn ``(batch, pixels)`` tensor instead -- the graph-captured training step keeps it in a static
    device buffer (``qiddm_amd.trainer``)."""
    if data.dim() == 1:
        data = data.unsqueeze(0)
    batch, pixels = data.shape
    if noise is None:
        noise = torch.normal(mean=0.5, std=0.2, size=(batch, pixels)).to(data.device)
    w = _schedule(tau, decay_mod, data.device)
    noisy = data.unsqueeze(1) * (1 - w) + noise.unsqueeze(1) * w        # (batch, tau, pixels)
    return noisy.clamp(0, 1).reshape(batch * tau, pixels)


def _draw_field(data):
    """The one N(0.5, 0.2) draw of ``add_normal_noise_multiple``, float32 on the CPU generator, moved to the
    data's device (src/noise.py:113-115)."""
    return torch.normal(mean=0.5, std=0.2, size=tuple(data.shape)).to(data.device)


# marks this schedule as the one the fused training step implements; ``noise_field(data)`` supplies its draw
add_normal_noise_multiple.noise_field = _draw_field
add_normal_noise_multiple.schedule = _schedule
