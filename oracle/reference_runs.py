"""Oracle restatement of the reference's ``test()`` pipeline for the shipped trajectories.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

``results_rebuttal_complex_dataset/{medmnist,logo2kplus}.zip`` in the reference hold, for five
label folders, a trained ``QIDDM_PL_noise(784, 8, 6, 2)`` checkpoint and the images
``image_{1..10}/step_{1..6}.png`` its authors' run of ``src/bloodmnist.py`` wrote:

  seed            ``torch.manual_seed(42)`` (``src/bloodmnist.py:374-377``, ``--seed`` default)
  first_x         ``torch.rand(10, 1, 28, 28, double) * 0.75 + 0.5`` (``:411``)
  sampling        ``diff.sample(first_x, n_iters=5, only_last=False)`` (``:231-233``) with
                  ``prediction_goal="data"``: ``x <- net(x)`` (``src/models.py:127-129``)
  net             ``QIDDM_PL_noise.forward`` (``nn/qdense.py:1424-1448``): PCA(8) re-fit on the batch,
                  two chained rounds of six [RZ(x_j) ; SEL(2 layers, CZ)] blocks, <Z_i>, ``linear_up``
  post            clamp(0,1) * 255 (``src/bloodmnist.py:236-240``) and ``plt.imsave(cmap="gray")``
                  per image and step (``:273-277``): per-image min-max normalisation, 256 grey levels

The fixtures (``tests/golden/reference_runs/``, written by ``tests/golden/make_reference_runs.py``) are
outputs of PennyLane-Lightning; reproducing them pins the RZ / Rot / SEL-range / CZ-ring / <Z> / wire-order
conventions of ``oracle.statevector`` and the PCA sign rule of ``oracle.pca``.
"""
from __future__ import annotations

import numpy as np
import torch

from . import circuits as oc
from . import pca as opca


def first_x(seed: int = 42, n: int = 10, size: int = 28) -> torch.Tensor:
    """``src/bloodmnist.py:374-377, 411``: the first torch draw after seeding."""
    torch.manual_seed(seed)
    return torch.rand(n, 1, size, size, dtype=torch.double) * 0.75 + 0.5


# matplotlib's 256-entry "gray" table is ``linspace(0, 1, 256)``; ``to_rgba(bytes=True)`` converts it with
# ``(lut * 255).astype(uint8)`` -- a truncation, so 24 of the 256 entries come out one level low (33, 37, ...).
_GRAY_LUT8 = (np.linspace(0.0, 1.0, 256) * 255).astype(np.uint8).astype(np.int64)


def imsave_gray_levels(img: np.ndarray) -> np.ndarray:
    """Grey level matplotlib's ``imsave(cmap="gray")`` writes for every pixel: ``Normalize(min, max)``, the
    256-entry colour map lookup ``floor(x * 256)`` clipped to 255, and the byte table above.  (Checked against
    matplotlib 3.10's own ``imsave``: identical on random images.)"""
    img = np.asarray(img, dtype=np.float64)
    lo, hi = img.min(), img.max()
    x = (img - lo) / (hi - lo) if hi > lo else np.zeros_like(img)
    return _GRAY_LUT8[np.clip(np.floor(x * 256.0), 0, 255).astype(np.int64)]


def imsave_gray_index(img: np.ndarray) -> np.ndarray:
    """The colour-map index (0..255) behind ``imsave_gray_levels``: the quantity a numerical difference moves by one
    step; the byte table then shows that step as 0, 1 or 2 grey levels (tests/test_oracle_reference_runs.py)."""
    img = np.asarray(img, dtype=np.float64)
    lo, hi = img.min(), img.max()
    x = (img - lo) / (hi - lo) if hi > lo else np.zeros_like(img)
    return np.clip(np.floor(x * 256.0), 0, 255).astype(np.int64)


# for every grey level a PNG can hold: the colour indices that produce it (one, or two where the byte table
# truncates the next entry down onto it; none for the 24 levels the table skips)
_INDEX_LO = np.full(256, 10 ** 6, dtype=np.int64)
_INDEX_HI = np.full(256, -(10 ** 6), dtype=np.int64)
for _i, _l in enumerate(_GRAY_LUT8):
    _INDEX_LO[_l] = min(_INDEX_LO[_l], _i)
    _INDEX_HI[_l] = max(_INDEX_HI[_l], _i)


def index_steps_from_levels(index: np.ndarray, ref_levels: np.ndarray) -> np.ndarray:
    """Distance, in colour-index steps, from ``index`` to the nearest index that the byte table maps to the saved
    grey level ``ref_levels`` (huge where the level cannot come out of the table at all)."""
    lo, hi = _INDEX_LO[ref_levels], _INDEX_HI[ref_levels]
    return np.where(index < lo, lo - index, np.where(index > hi, index - hi, 0))


def indices_from_images(stack: torch.Tensor) -> np.ndarray:
    """``levels_from_images`` before the byte table: ``(batch, iters, H, W)`` colour indices."""
    out = torch.clamp(torch.clamp(stack.double().cpu(), 0.0, 1.0) * 255.0, 0.0, 255.0).numpy()
    it, b = out.shape[:2]
    ix = np.zeros((b, it) + out.shape[3:], dtype=np.int64)
    for i in range(b):
        for s in range(it):
            ix[i, s] = imsave_gray_index(out[s, i, 0])
    return ix


def qiddm_pl_forward(x_img, weights1, up_w, up_b, pca=opca.pca_fit_transform):
    """``QIDDM_PL_noise.forward`` (``nn/qdense.py:1424-1448``) including the per-call PCA fit."""
    b = x_img.shape[0]
    n = weights1.shape[3]
    xr = torch.from_numpy(np.ascontiguousarray(pca(x_img.reshape(b, -1).numpy(), n)))
    ev = oc.run_circuit(oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz"), xr, weights1)
    return (ev @ up_w.T + up_b).reshape(x_img.shape)


def sample_levels(state_dict, n_iters: int = 5, seed: int = 42, forward=qiddm_pl_forward) -> np.ndarray:
    """The ``(10, n_iters + 1, 28, 28)`` grey levels ``test()`` saves for one checkpoint."""
    x = first_x(seed)
    imgs = [x]
    for _ in range(n_iters):
        x = forward(x, state_dict["weights1"], state_dict["linear_up.weight"], state_dict["linear_up.bias"])
        imgs.append(x)
    return levels_from_images(torch.stack(imgs))


def levels_from_images(stack: torch.Tensor) -> np.ndarray:
    """``(iters, batch, 1, H, W)`` images -> ``(batch, iters, H, W)`` saved grey levels."""
    out = torch.clamp(torch.clamp(stack.double().cpu(), 0.0, 1.0) * 255.0, 0.0, 255.0).numpy()
    it, b = out.shape[:2]
    lv = np.zeros((b, it) + out.shape[3:], dtype=np.int64)
    for i in range(b):
        for s in range(it):
            lv[i, s] = imsave_gray_levels(out[s, i, 0])
    return lv
