"""PCA front-end on the device (finding F4).

The PCA nets of the reference call ``sklearn.decomposition.PCA(n).fit_transform`` on every forward
(nn/qdense.py:456, 1429, 1722): device -> host copy, LAPACK (or the randomized solver, which the ``auto`` rule of
scikit-learn picks for (256, 784) batches and which draws from numpy's global RNG), host -> device copy.
``DevicePCA`` is the exact decomposition as torch expressions wherever the batch lives: centre, eigendecomposition
of the smaller of the two Gram matrices, scikit-learn's sign convention (``svd_flip(u_based_decision=False)``: the
largest-magnitude entry of every component is positive), scores ``U S``.  Same attribute names as the fitted
sklearn object for what the nets touch (``components_``, ``mean_``, ``singular_values_``, ``explained_variance_``,
``explained_variance_ratio_``, ``n_components_``).

It is opt-in (``use_device_pca(net)``): the default stays sklearn so that a seeded run reproduces the reference's
numbers including the randomized solver's approximation.
"""
from __future__ import annotations

import torch


class DevicePCA:
    def __init__(self, n_components: int):
        self.n_components = int(n_components)

    # -- fitting ---------------------------------------------------------------------------------------
    def _fit(self, x: torch.Tensor):
        if x.dim() != 2:
            raise ValueError(f"Expected 2D array, got {x.dim()}D array instead")
        n, f = x.shape
        k = self.n_components
        if not 0 <= k <= min(n, f):
            raise ValueError(f"n_components={k!r} must be between 0 and min(n_samples, n_features)={min(n, f)!r} "
                             "with svd_solver='full'")
        x = x.detach().to(torch.float64)
        mean = x.mean(dim=0)
        xc = x - mean
        if f <= n:
            evals, evecs = torch.linalg.eigh(xc.T @ xc)                  # ascending
            evals, evecs = evals.flip(0).clamp_min(0), evecs.flip(1)
            s = evals.sqrt()
            vt = evecs.T[:k]
            u = None
        else:
            evals, u_all = torch.linalg.eigh(xc @ xc.T)
            evals, u_all = evals.flip(0).clamp_min(0), u_all.flip(1)
            s = evals.sqrt()
            u = u_all[:, :k]
            vt = (u.T @ xc) / s[:k].clamp_min(torch.finfo(torch.float64).tiny).unsqueeze(1)
        # svd_flip(u_based_decision=False): sign of the largest-magnitude entry of each row of Vt
        idx = vt.abs().argmax(dim=1)
        signs = torch.sign(vt[torch.arange(k, device=vt.device), idx])
        signs = torch.where(signs == 0, torch.ones_like(signs), signs)
        vt = vt * signs.unsqueeze(1)
        scores = (xc @ vt.T) if u is None else u * signs * s[:k]
        total_var = (s ** 2).sum() / max(n - 1, 1)
        self.mean_, self.components_ = mean, vt.contiguous()
        self.singular_values_ = s[:k].clone()
        self.explained_variance_ = s[:k] ** 2 / max(n - 1, 1)
        self.explained_variance_ratio_ = self.explained_variance_ / total_var
        self.n_components_, self.n_samples_, self.n_features_in_ = k, n, f
        return scores

    def fit(self, x):
        self._fit(x)
        return self

    def fit_transform(self, x):
        return self._fit(x)

    def transform(self, x):
        return (x.to(torch.float64) - self.mean_) @ self.components_.T

    def inverse_transform(self, z):
        return z.to(torch.float64) @ self.components_ + self.mean_


# ---- what the layer classes call: one code path for both back-ends ----------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def fit_transform(pca, flat: torch.Tensor) -> torch.Tensor:
    if isinstance(pca, DevicePCA):
        return pca.fit_transform(flat)
    return torch.tensor(pca.fit_transform(_np(flat)))


def fit(pca, flat: torch.Tensor):
    pca.fit(flat) if isinstance(pca, DevicePCA) else pca.fit(_np(flat))
    return pca


def transform(pca, flat: torch.Tensor) -> torch.Tensor:
    if isinstance(pca, DevicePCA):
        return pca.transform(flat)
    return torch.tensor(pca.transform(_np(flat)))


def inverse_transform(pca, z: torch.Tensor) -> torch.Tensor:
    if isinstance(pca, DevicePCA):
        return pca.inverse_transform(z)
    return torch.tensor(pca.inverse_transform(_np(z)))


def use_device_pca(net: torch.nn.Module, enable: bool = True) -> torch.nn.Module:
    """Swap every ``.pca`` of ``net`` (and its sub-modules) between sklearn's PCA and :class:`DevicePCA` with the
    same number of components.  Fitted state is dropped: these nets re-fit on every forward anyway (F4)."""
    from sklearn.decomposition import PCA
    for m in net.modules():
        p = getattr(m, "pca", None)
        if p is None:
            continue
        k = p.n_components
        if enable and not isinstance(p, DevicePCA):
            m.pca = DevicePCA(k)
        elif not enable and isinstance(p, DevicePCA):
            m.pca = PCA(n_components=k)
    return net
