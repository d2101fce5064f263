"""DevicePCA (torch, any device) == sklearn.decomposition.PCA with an exact solver (finding F4; reference
nn/qdense.py:456, 1429, 1722 call ``PCA(n).fit_transform`` per forward)."""
import numpy as np
import pytest
import torch
from sklearn.decomposition import PCA

from qiddm_amd.pca import DevicePCA, use_device_pca


def _check(device, n, f, k, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.random((n, f)) @ rng.random((f, f))            # correlated columns
    ref = PCA(n_components=k, svd_solver="full")
    want = ref.fit_transform(x)
    xt = torch.tensor(x, device=device)
    pca = DevicePCA(k)
    got = pca.fit_transform(xt)
    scale = np.abs(want).max()
    assert np.abs(got.cpu().numpy() - want).max() < 1e-9 * scale
    assert np.abs(pca.components_.cpu().numpy() - ref.components_).max() < 1e-8
    assert np.allclose(pca.mean_.cpu().numpy(), ref.mean_, atol=1e-12)
    assert np.allclose(pca.singular_values_.cpu().numpy(), ref.singular_values_, rtol=1e-9)
    assert np.allclose(pca.explained_variance_.cpu().numpy(), ref.explained_variance_, rtol=1e-9)
    assert np.allclose(pca.explained_variance_ratio_.cpu().numpy(), ref.explained_variance_ratio_, rtol=1e-9)
    y = rng.random((5, f))
    assert np.abs(pca.transform(torch.tensor(y, device=device)).cpu().numpy() - ref.transform(y)).max() < 1e-9 * scale
    z = rng.random((5, k))
    assert np.allclose(pca.inverse_transform(torch.tensor(z, device=device)).cpu().numpy(), ref.inverse_transform(z),
                       atol=1e-9 * scale)


@pytest.mark.parametrize("n,f,k", [(40, 16, 4), (12, 64, 6), (256, 784, 8), (30, 30, 10)])
def test_device_pca_matches_sklearn_full_cpu(n, f, k):
    _check("cpu", n, f, k)


@pytest.mark.gpu
@pytest.mark.parametrize("n,f,k", [(64, 64, 6), (256, 784, 10)])
def test_device_pca_matches_sklearn_full_gpu(n, f, k):
    _check("cuda", n, f, k)


def test_too_few_rows_raises_like_sklearn():
    with pytest.raises(ValueError, match="n_components=8 must be between 0 and min"):
        DevicePCA(8).fit_transform(torch.rand(4, 64, dtype=torch.float64))
    with pytest.raises(ValueError):
        PCA(n_components=8, svd_solver="full").fit_transform(np.random.rand(4, 64))


@pytest.mark.gpu
def test_pca_nets_run_fully_on_device():
    """differN_noise / QIDDM_PL_noise with the device front-end: no host round trip, same post-PCA pipeline."""
    from qiddm_amd import nn
    torch.manual_seed(0)
    x = torch.rand(24, 1, 8, 8, dtype=torch.float64, device="cuda")
    for net in (nn.differN_noise(8, 2, 2).to("cuda"), nn.QIDDM_PL_noise(64, 4, 2, 1).to("cuda")):
        ref_pca = PCA(n_components=net.pca.n_components, svd_solver="full")
        red_ref = torch.tensor(ref_pca.fit_transform(x.reshape(24, -1).cpu().numpy()))
        use_device_pca(net)
        assert isinstance(net.pca, DevicePCA)
        with torch.no_grad():
            got = net(x)
            if hasattr(net, "forward_from_reduced"):
                want = net.forward_from_reduced(red_ref.to(torch.float32).to("cuda"))
            else:
                ev = net.quantum_rounds(red_ref.to("cuda").to(net.linear_up.weight.dtype))
                want = net.linear_up(ev.to(net.linear_up.weight.dtype)).view(24, 1, 8, 8)
        assert got.shape == x.shape and torch.allclose(got.double(), want.double(), atol=2e-4)
        use_device_pca(net, False)
        assert isinstance(net.pca, PCA)
