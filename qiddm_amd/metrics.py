"""Image-quality metrics of the reference's evaluation (src/metrics.py:162-356), on the device.

The reference moves every (generated, real) pair to the CPU and calls skimage / sklearn / scipy once per
pair inside three nested Python loops.  Here each metric is one batched tensor expression over all
(iteration, generated, real) triples, float64, wherever the images live:

* ``cosine_similarity``  ``0.5 + 0.5 cos`` (``calculate_cos`` :162-173, ``get_cosine_similarity`` :176-209)
* ``ssim``               skimage ``structural_similarity`` defaults -- 7x7 uniform window, sample covariance,
                          K1 = 0.01, K2 = 0.03, borders cropped -- with the reference's ``data_range =
                          generated.max() - generated.min()`` per generated image (:212-247)
* ``psnr``               skimage ``peak_signal_noise_ratio(real, generated, data_range=...)`` (:276-309)
* ``fid``                the pixel-space Frechet distance of ``calculate_fid`` (:345-355): no Inception network

Inputs as the reference passes them: ``generated`` (iterations, G, 1, H, W), ``real`` (R, 1, H, W); each
function returns the per-iteration list the reference plots (mean over all pairs).  The plotting helpers
(matplotlib) are out of scope.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _prep(generated, real, gen_img_count=None, real_img_count=None):
    if gen_img_count is not None and gen_img_count < generated.shape[1]:
        generated = generated[:, :gen_img_count]
    if real_img_count is not None and real_img_count < real.shape[0]:
        real = real[:real_img_count]
    g = generated.to(torch.float64)
    r = real.to(device=g.device, dtype=torch.float64)
    return g.reshape(g.shape[0], g.shape[1], g.shape[-2], g.shape[-1]), r.reshape(r.shape[0], r.shape[-2], r.shape[-1])


def cosine_similarity(generated, real, gen_img_count=None, real_img_count=None):
    g, r = _prep(generated, real, gen_img_count, real_img_count)
    gv, rv = g.flatten(2), r.flatten(1)                                   # (I, G, P), (R, P)
    num = gv @ rv.T                                                       # (I, G, R)
    den = gv.norm(dim=2, keepdim=True) * rv.norm(dim=1)
    res = num / den
    res = torch.where(torch.isneginf(res), torch.zeros_like(res), res)    # reference :172
    return (0.5 + 0.5 * res).mean(dim=(1, 2)).tolist()


def _data_range(g):
    flat = g.flatten(2)
    return (flat.max(dim=2).values - flat.min(dim=2).values)              # (I, G)


def ssim_pairs(g, r, win_size=7, k1=0.01, k2=0.03):
    """(I, G, R) SSIM of every generated / real pair."""
    i, gn, h, w = g.shape
    if min(h, w) < win_size:
        raise ValueError("win_size exceeds image extent.")                # skimage's message
    npix = win_size * win_size
    cov_norm = npix / (npix - 1.0)                                        # use_sample_covariance=True

    def box(t):                                                           # valid windows == crop of the reflect filter
        return F.avg_pool2d(t.unsqueeze(1), win_size, stride=1).squeeze(1)

    gf = g.reshape(i * gn, h, w)
    ux, uxx = box(gf), box(gf * gf)                                       # (IG, h', w')
    uy, uyy = box(r), box(r * r)                                          # (R, h', w')
    uxy = box((gf.unsqueeze(1) * r.unsqueeze(0)).reshape(-1, h, w)).reshape(i * gn, r.shape[0], *ux.shape[1:])
    ux, uxx = ux.unsqueeze(1), uxx.unsqueeze(1)
    uy, uyy = uy.unsqueeze(0), uyy.unsqueeze(0)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    rng = _data_range(g).reshape(i * gn, 1, 1, 1)
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    return s.mean(dim=(2, 3)).reshape(i, gn, r.shape[0])


def ssim(generated, real, gen_img_count=None, real_img_count=None):
    g, r = _prep(generated, real, gen_img_count, real_img_count)
    return ssim_pairs(g, r).mean(dim=(1, 2)).tolist()


def ssim_single(generated, real, gen_img_count=None, real_img_count=None):
    """``get_ssim_single`` (:250-272): against the first real image only."""
    g, r = _prep(generated, real, gen_img_count, real_img_count)
    return ssim_pairs(g, r[:1]).mean(dim=(1, 2)).tolist()


def psnr(generated, real, gen_img_count=None, real_img_count=None):
    g, r = _prep(generated, real, gen_img_count, real_img_count)
    err = ((r.unsqueeze(0).unsqueeze(0) - g.unsqueeze(2)) ** 2).mean(dim=(3, 4))      # (I, G, R)
    rng = _data_range(g).unsqueeze(2)
    return (10.0 * torch.log10(rng * rng / err)).mean(dim=(1, 2)).tolist()


def _trace_sqrt_product(s1, s2):
    """tr sqrtm(s1 s2) for symmetric PSD s1, s2 = sum of sqrt of the eigenvalues of s1^(1/2) s2 s1^(1/2)."""
    w, v = torch.linalg.eigh(s1)
    root = (v * w.clamp_min(0).sqrt()) @ v.T
    ev = torch.linalg.eigvalsh(root @ s2 @ root)
    return ev.clamp_min(0).sqrt().sum()


def frechet_distance(act1, act2):
    """``calculate_fid`` (:345-355) on (n1, P) and (n2, P) float64 activations (here: raw pixels)."""
    mu1, mu2 = act1.mean(dim=0), act2.mean(dim=0)
    s1, s2 = torch.cov(act1.T), torch.cov(act2.T)
    ssdiff = ((mu1 - mu2) ** 2).sum()
    return ssdiff + torch.trace(s1) + torch.trace(s2) - 2.0 * _trace_sqrt_product(s1, s2)


def fid(generated, real, gen_img_count=None, real_img_count=None):
    g, r = _prep(generated, real, gen_img_count, real_img_count)
    rv = r.flatten(1)
    return [frechet_distance(g[i].flatten(1), rv).item() for i in range(g.shape[0])]
