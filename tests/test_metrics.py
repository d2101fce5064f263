"""Batched on-device metrics (qiddm_amd.metrics) against the oracle's per-pair loops (reference
src/metrics.py:162-356) and their known answers."""
import numpy as np
import pytest
import torch

from oracle import metrics as om
from qiddm_amd import metrics as qm


def _data(seed=0, iters=3, g=4, r=5, side=12):
    rng = np.random.default_rng(seed)
    gen = rng.random((iters, g, 1, side, side))
    real = rng.random((r, 1, side, side))
    gen[1] *= 0.5                                     # different data ranges per iteration
    return gen, real


def _check(device):
    gen, real = _data()
    tg, tr = torch.tensor(gen, device=device), torch.tensor(real, device=device)
    assert qm.cosine_similarity(tg, tr) == pytest.approx(om.per_iteration(gen, real, "cos"), rel=1e-12)
    assert qm.ssim(tg, tr) == pytest.approx(om.per_iteration(gen, real, "ssim"), rel=1e-10)
    assert qm.psnr(tg, tr) == pytest.approx(om.per_iteration(gen, real, "psnr"), rel=1e-12)
    assert qm.fid(tg, tr) == pytest.approx(om.per_iteration(gen, real, "fid"), rel=1e-7, abs=1e-8)
    assert qm.ssim_single(tg, tr) == pytest.approx(om.per_iteration(gen, real[:1], "ssim"), rel=1e-10)
    # image-count truncation as the reference applies it
    assert qm.psnr(tg, tr, gen_img_count=2, real_img_count=3) == \
        pytest.approx(om.per_iteration(gen[:, :2], real[:3], "psnr"), rel=1e-12)


def test_metrics_cpu_vs_oracle():
    _check("cpu")


@pytest.mark.gpu
def test_metrics_gpu_vs_oracle():
    _check("cuda")


def test_known_answers():
    x = torch.rand(1, 3, 1, 10, 10, dtype=torch.float64)
    same = x[0]
    s = qm.ssim_pairs(x.reshape(1, 3, 10, 10), same.reshape(3, 10, 10))
    assert torch.allclose(torch.diagonal(s[0]), torch.ones(3, dtype=torch.float64), atol=1e-12)   # SSIM(x, x) = 1
    assert qm.cosine_similarity(x, same)[0] <= 1.0
    a = torch.rand(6, 20, dtype=torch.float64)
    assert abs(qm.frechet_distance(a, a).item()) < 1e-6                                            # FID(x, x) = 0
    shifted = a + 2.0
    assert qm.frechet_distance(a, shifted).item() == pytest.approx(20 * 4.0, rel=1e-9)             # pure mean shift
    with pytest.raises(ValueError):
        qm.ssim(torch.rand(1, 1, 1, 5, 5), torch.rand(1, 1, 5, 5))                                 # window > image
    z = torch.zeros(1, 1, 1, 8, 8, dtype=torch.float64)
    z[0, 0, 0, 0, 0] = 1.0
    assert qm.psnr(z, z[0])[0] == float("inf")                                                     # identical images
