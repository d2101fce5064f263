"""Oracle restatement of the reference's image metrics (src/metrics.py:162-356) with numpy / scipy.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The reference delegates to scikit-image
(``structural_similarity``, ``peak_signal_noise_ratio``), which is not installed here: **parity unpinned** for
SSIM / PSNR beyond their published definitions (Wang et al. 2004; skimage 0.19 defaults: 7x7 uniform filter,
sample covariance, K1 0.01, K2 0.03, border crop).  ``calculate_cos`` and ``calculate_fid`` are numpy / scipy in
the reference itself and are restated line by line (scipy.linalg.sqrtm is available)."""
from __future__ import annotations

import numpy as np
from scipy.linalg import sqrtm
from scipy.ndimage import uniform_filter


def calculate_cos(v1, v2):
    """src/metrics.py:162-173; v1 (1,H,W) or (H,W), v2 (1,H,W)."""
    pixels = v2.shape[-1] * v2.shape[-2]
    a, b = np.asarray(v1).reshape(-1, pixels), np.asarray(v2).reshape(-1, pixels)
    res = a @ b.T / (np.linalg.norm(a, axis=1).reshape(-1, 1) * np.linalg.norm(b, axis=1))
    res[np.isneginf(res)] = 0
    return 0.5 + 0.5 * res


def structural_similarity(im1, im2, data_range, win_size=7, k1=0.01, k2=0.03):
    im1, im2 = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    npix = win_size ** 2
    cov_norm = npix / (npix - 1)
    ux, uy = uniform_filter(im1, size=win_size), uniform_filter(im2, size=win_size)
    uxx, uyy, uxy = (uniform_filter(im1 * im1, size=win_size), uniform_filter(im2 * im2, size=win_size),
                     uniform_filter(im1 * im2, size=win_size))
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return s[pad:-pad, pad:-pad].mean()


def peak_signal_noise_ratio(image_true, image_test, data_range):
    err = np.mean((np.asarray(image_true, dtype=np.float64) - np.asarray(image_test, dtype=np.float64)) ** 2)
    return 10 * np.log10(data_range ** 2 / err)


def calculate_fid(act1, act2, n1, n2):
    """src/metrics.py:345-355."""
    act1, act2 = np.asarray(act1).reshape([n1, -1]), np.asarray(act2).reshape([n2, -1])
    mu1, sigma1 = act1.mean(axis=0), np.cov(act1, rowvar=False)
    mu2, sigma2 = act2.mean(axis=0), np.cov(act2, rowvar=False)
    covmean = sqrtm(sigma1.dot(sigma2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return np.sum((mu1 - mu2) ** 2.0) + np.trace(sigma1 + sigma2 - 2.0 * covmean)


def per_iteration(generated, real, kind):
    """The triple loops of get_cosine_similarity / get_ssim / get_psnr / get_fid (:176-342)."""
    out = []
    for it in range(generated.shape[0]):
        vals = []
        if kind == "fid":
            vals.append(calculate_fid(generated[it], real, generated.shape[1], real.shape[0]))
        else:
            for i in range(generated.shape[1]):
                for j in range(real.shape[0]):
                    g, r = generated[it, i].squeeze(), real[j].squeeze()
                    rng = g.max() - g.min()
                    if kind == "cos":
                        vals.append(float(calculate_cos(generated[it, i], real[j]).reshape(-1)[0]))
                    elif kind == "ssim":
                        vals.append(structural_similarity(g, r, rng))
                    else:
                        vals.append(peak_signal_noise_ratio(r, g, rng))
        out.append(float(np.mean(vals)))
    return out
