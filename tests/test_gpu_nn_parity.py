"""GPU parity of the layer classes (the reference's ``nn`` namespace) and of the denoise loop
against the CPU oracle restatement of the same reference lines, on identical seeds."""
import os

import pytest
import torch

from oracle import circuits as oc
from oracle import diffusion as odf

pytestmark = pytest.mark.gpu

CK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoints")
DEV = "cuda"


def _img(b, w, seed, c=1):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(b, c, w, w, generator=g, dtype=torch.float64)


@pytest.mark.parametrize("cfg", [(64, 4, 2, 8), (784, 8, 14, 28)])
def test_qnn_noise_forward(cfg):
    """Row A1.  C1: QNN_noise(64,4,2); C2: QNN_noise(784,8,14)."""
    from qiddm_amd import nn
    dim, n, depth, w = cfg
    torch.manual_seed(42)
    m = nn.QNN_noise(dim, n, depth).to(DEV)
    x = _img(33, w, 1)
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = oc.qnn_forward(x, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights"],
                         sd["linear_up.weight"], sd["linear_up.bias"])
    assert got.shape == x.shape and got.dtype == torch.float64
    assert torch.allclose(got, ref, atol=5e-5, rtol=1e-4), (got - ref).abs().max()


def test_qnn_noise_real_checkpoint():
    """Trained weights shipped with the reference (results/emnist.zip)."""
    from qiddm_amd import models, nn, noise
    ck = torch.load(os.path.join(CK, "QNN_linear_features=8_qdepth=6_add_noise=0_noise_2.pt"),
                    weights_only=True, map_location="cpu")
    net = nn.QNN_noise(784, 8, 6)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "noise", (28, 28)).to(DEV, dtype=torch.double)
    diff.load_state_dict(ck["model_state_dict"])
    x = _img(10, 28, 3) * 0.75 + 0.5
    sd = {k[4:]: v for k, v in ck["model_state_dict"].items()}
    ref_net = lambda t: oc.qnn_forward(t, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights"],
                                       sd["linear_up.weight"], sd["linear_up.bias"])
    diff.eval()
    got = diff.sample(first_x=x.to(DEV), n_iters=3).cpu()
    ref = odf.sample(ref_net, x, 3, goal="noise")
    assert got.shape == (4 * 28, 10 * 28)
    assert torch.allclose(got, ref, atol=1e-4), (got - ref).abs().max()


@pytest.mark.parametrize("cfg,fused", [((64, 4, 2, 1, 8), True), ((784, 8, 6, 2, 28), True),
                                        ((784, 6, 14, 2, 28), False)])
def test_qiddm_ll_noise_forward(cfg, fused):
    """Row A2: QIDDM_LL_noise; fused multi-round launch (no_grad) and the per-round QNode path."""
    from qiddm_amd import nn
    dim, n, L, N, w = cfg
    torch.manual_seed(7)
    m = nn.QIDDM_LL_noise(dim, n, L, N).to(DEV, dtype=torch.double)
    x = _img(21, w, 2)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = oc.qiddm_ll_forward(x, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights1"],
                              sd["linear_up.weight"], sd["linear_up.bias"])
    if fused:
        with torch.no_grad():
            got = m(x.to(DEV)).cpu()
    else:
        got = m(x.to(DEV)).detach().cpu()
    assert torch.allclose(got, ref, atol=5e-5, rtol=1e-4), (got - ref).abs().max()


def test_detach_quantum_switch():
    """F1: as written only linear_up learns; detach_quantum=False lets parameter-shift
    gradients reach weights1 and linear_down -- and they match autograd through the oracle."""
    from qiddm_amd import nn, set_default_precision
    torch.manual_seed(3)
    x = _img(6, 8, 4)
    m = nn.QIDDM_LL_noise(64, 4, 2, 2).to(DEV, dtype=torch.double)
    m(x.to(DEV)).square().mean().backward()
    assert m.weights1.grad is None and m.linear_down.weight.grad is None
    assert m.linear_up.weight.grad is not None
    set_default_precision("f64")
    try:
        m2 = nn.QIDDM_LL_noise(64, 4, 2, 2, detach_quantum=False).to(DEV, dtype=torch.double)
        m2.load_state_dict(m.state_dict())
        m2(x.to(DEV)).square().mean().backward()
    finally:
        set_default_precision("f32")
    ps = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.named_parameters()}
    ref = oc.qiddm_ll_forward(x, ps["linear_down.weight"], ps["linear_down.bias"], ps["weights1"],
                              ps["linear_up.weight"], ps["linear_up.bias"])
    ref.square().mean().backward()
    for k, p in m2.named_parameters():
        assert torch.allclose(p.grad.cpu(), ps[k].grad, atol=1e-9), k


def test_differn_from_reduced_real_checkpoint():
    """Row A3 from the post-PCA tensor on (F4), with the Ray-Tune checkpoint's weights."""
    from qiddm_amd import nn
    ck = torch.load(os.path.join(CK, "differN_noise=9_N=2_w28_h28_noise_0.035069821502010365_0.25081669882500224.pt"),
                    weights_only=True, map_location="cpu")
    m = nn.differN_noise_befor(28, 9, 2)
    m.load_state_dict({"weights": ck["model_state_dict"]["net.weights"]})
    m = m.to(DEV)
    g = torch.Generator().manual_seed(11)
    red = torch.randn(17, 10, generator=g) * 2.0
    ref = oc.differn_from_reduced(red, ck["model_state_dict"]["net.weights"], (28, 28))
    with torch.no_grad():
        fused = m.forward_from_reduced(red.to(DEV)).cpu()
    per_round = m.forward_from_reduced(red.to(DEV)).detach().cpu()
    # post-processed pixels are probabilities * 784: scale the tolerance accordingly
    assert torch.allclose(fused, ref, atol=2e-2), (fused - ref).abs().max()
    assert torch.allclose(per_round, ref, atol=2e-2), (per_round - ref).abs().max()
    assert fused.shape == (17, 1, 28, 28)


def test_differn_full_forward_with_host_pca():
    from qiddm_amd import nn
    torch.manual_seed(0)
    m = nn.differN_noise(8, 3, 2).to(DEV)
    x = _img(20, 8, 5)
    with torch.no_grad():
        y = m(x.to(DEV))
    red = m.reduce(x)
    ref = oc.differn_from_reduced(red.cpu(), m.weights.detach().cpu(), (8, 8))
    assert torch.allclose(y.cpu(), ref, atol=2e-3)
    with pytest.raises(ValueError):      # fewer rows than PCA components (sklearn, F4)
        m(x[:3].to(DEV))


@pytest.mark.parametrize("cls_name,wmap", [("QDenseUndirected_old", "qw_tanh"), ("QDenseUndirected_old_noise", "tanh")])
def test_qdense_undirected_forward(cls_name, wmap):
    """Row A4 at C1 size (8x8, n=6) and the shipped 28x28 / 60-layer checkpoint."""
    from qiddm_amd import nn
    torch.manual_seed(5)
    m = getattr(nn, cls_name)(12, 8).to(DEV)
    x = _img(9, 8, 6)
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    ref = oc.qdense_undirected_forward(x, m.weights.detach().cpu(), (8, 8), wmap)
    assert torch.allclose(got, ref, atol=2e-3), (got - ref).abs().max()


def test_qdense_real_checkpoint_28():
    from qiddm_amd import nn
    ck = torch.load(os.path.join(CK, "QDenseUndirected_old_noise60_w28_h28_noise0_noise_2.pt"),
                    weights_only=True, map_location="cpu")
    m = nn.QDenseUndirected_old_noise(60, 28)
    m.load_state_dict({"weights": ck["model_state_dict"]["net.weights"]})
    m = m.to(DEV)
    x = _img(5, 28, 7)
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    ref = oc.qdense_undirected_forward(x, ck["model_state_dict"]["net.weights"], (28, 28), "tanh")
    assert torch.allclose(got, ref, atol=2e-2), (got - ref).abs().max()


@pytest.mark.parametrize("cin,cout,k,pad,hw", [(1, 8, 3, 1, 9), (8, 16, 3, 1, 6), (16, 8, 1, 0, 7), (3, 4, 3, 1, 5)])
def test_qconv2d_forward(cin, cout, k, pad, hw):
    """Row A5: the intended QConv2d (F3)."""
    from qiddm_amd import nn
    torch.manual_seed(8)
    m = nn.QConv2d(cin, cout, k, pad, 3).to(DEV)
    x = _img(3, hw, 9, c=cin)
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    ref = oc.qconv2d_forward(x, m.weights.detach().cpu(), cout, (k, k), (pad, pad))
    assert got.shape == ref.shape == (3, cout, hw, hw)
    assert torch.allclose(got, ref, atol=1e-3), (got - ref).abs().max()


def test_unet_simple_runs():
    """Row A6: UNetUndirectedS(3, 8, 3) end to end (wires 4,7,8 / 5,9 / 4,8)."""
    from qiddm_amd import nn
    torch.manual_seed(9)
    # the drivers cast the whole model to double (src/mnist_exm.py:449); BatchNorm2d is float32 otherwise
    u = nn.UNetUndirectedS(3, 8, 3).to(DEV, dtype=torch.double).eval()
    with torch.no_grad():
        y = u(_img(2, 28, 10).to(DEV))
    assert y.shape == (2, 1, 28, 28) and y.dtype == torch.float64 and torch.isfinite(y).all()


def test_diffusion_training_step_on_device():
    """Row A7 with a quantum net: loss + recon of one training step vs the oracle loop."""
    from qiddm_amd import models, nn, noise
    torch.manual_seed(12)
    net = nn.QNN_noise(64, 4, 2)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8),
                            torch.nn.MSELoss()).to(DEV, dtype=torch.double)
    diff.train()
    x = _img(4, 8, 13).reshape(4, 64)
    torch.manual_seed(77)
    nz = torch.normal(mean=0.5, std=0.2, size=(4, 64))
    torch.manual_seed(77)
    loss, recon = diff(x=x.to(DEV), T=10, verbose=True)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref_net = lambda t: oc.qnn_forward(t, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights"],
                                       sd["linear_up.weight"], sd["linear_up.bias"])
    ref_loss, ref_recon = odf.training_loss(ref_net, x, 10, (8, 8), "data", noise=nz)
    assert recon.shape == (40, 1, 8, 8)
    assert torch.allclose(recon.detach().cpu(), ref_recon.abs(), atol=5e-5)
    assert abs(loss.item() - ref_loss.item()) < 1e-5
    assert net.linear_up.weight.grad is not None and net.weights.grad is None   # F1


@pytest.mark.parametrize("precision,tol", [("f32", 5e-5), ("f64", 1e-10)])
@pytest.mark.parametrize("n,N,L,S,P", [(8, 1, 1, 14, 784), (8, 2, 6, 2, 784), (4, 1, 2, 2, 64), (2, 2, 1, 3, 10),
                                       (6, 2, 14, 2, 784), (10, 1, 3, 2, 100),
                                       # every thread-bit layout of the four-wave sampler: spare lane bits (3, 5), one wave
                                       # per replica (7), one and two register bits (9, 10); a one-layer round (generated
                                       # product state measured directly)
                                       (3, 1, 2, 2, 9), (5, 2, 1, 4, 25), (7, 1, 3, 1, 49), (9, 2, 2, 2, 81),
                                       (8, 2, 1, 1, 64)])
def test_dense_forward_kernel(n, N, L, S, P, precision, tol):
    """qiddm_dense_forward: linear_down -> rounds -> linear_up in one launch, both post modes,
    including the sub-wave layouts (n < 6: several samples per wavefront) and a ragged batch."""
    from qiddm_amd.circuit import Circuit, dense_forward
    g = torch.Generator().manual_seed(n * 100 + P)
    B = 45
    x = torch.rand(B, P, generator=g, dtype=torch.float64)
    wd = torch.randn(n, P, generator=g, dtype=torch.float64) / P ** 0.5 * 3
    bd = torch.randn(n, generator=g, dtype=torch.float64)
    wu = torch.randn(P, n, generator=g, dtype=torch.float64)
    bu = torch.randn(P, generator=g, dtype=torch.float64) * 0.1
    w = torch.randn(N, L, S, n, 3, generator=g, dtype=torch.float64) * 0.6
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=N, n_blocks=L,
                   sel_layers=S)
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz")
    ref = oc.run_circuit(spec, x @ wd.T + bd, w) @ wu.T + bu
    dev = lambda t: t.to(DEV)
    got = dense_forward(circ, dev(x), dev(wd), dev(bd), dev(w), dev(wu), dev(bu), precision).cpu()
    assert torch.allclose(got, ref, atol=tol, rtol=tol), (got - ref).abs().max()
    got1 = dense_forward(circ, dev(x), dev(wd), dev(bd), dev(w), dev(wu), dev(bu), precision,
                         post_mode=1, noise_factor=0.7).cpu()
    ref1 = torch.clamp(x - (ref - 0.5) * 0.1 * 0.7, 0, 1)
    assert torch.allclose(got1, ref1, atol=tol, rtol=tol), (got1 - ref1).abs().max()
    # no biases
    got2 = dense_forward(circ, dev(x), dev(wd), None, dev(w), dev(wu), None, precision).cpu()
    ref2 = oc.run_circuit(spec, x @ wd.T, w) @ wu.T
    assert torch.allclose(got2, ref2, atol=tol, rtol=tol)


def test_c5_style_16_qubit_qdense():
    """BASELINE config 5 shape: 28x28x3 = 2352 pixels, 16-qubit LL-style net (2352, 16, 6, 2):
    linear glue on torch, circuit on the n > 10 tiled kernel."""
    from qiddm_amd import nn
    torch.manual_seed(21)
    m = nn.QIDDM_LL_noise(2352, 16, 6, 2).to(DEV, dtype=torch.double)
    x = torch.rand(3, 1, 28, 84, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    ref = oc.qiddm_ll_forward(x, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights1"],
                              sd["linear_up.weight"], sd["linear_up.bias"])
    assert got.shape == x.shape
    assert torch.allclose(got, ref, atol=5e-5, rtol=1e-4), (got - ref).abs().max()


def test_c4_style_12_qubit_qconv():
    """BASELINE config 4 shape: QConv2d(C_in=256, C_out=256, k=3) -> 12 wires (nn/qconv.py:24-28)."""
    import warnings
    from qiddm_amd import nn
    torch.manual_seed(22)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = nn.QConv2d(256, 256, 3, 1, 3).to(DEV)
    assert m.wires == 12
    x = torch.rand(1, 256, 3, 3, dtype=torch.float64)
    with torch.no_grad():
        got = m(x.to(DEV)).cpu()
    ref = oc.qconv2d_forward(x, m.weights.detach().cpu(), 256, (3, 3), (1, 1))
    assert got.shape == (1, 256, 3, 3)
    assert torch.allclose(got, ref, atol=1e-3), (got - ref).abs().max()


@pytest.mark.parametrize("cin,cout,k,pad,hw", [(1, 8, 3, 1, 9), (8, 16, 3, 1, 6), (32, 16, 1, 0, 7), (16, 8, 3, 1, 11),
                                               (2, 2, (2, 3), (1, 0), 6)])
def test_qconv2d_fused_equals_traced_path(cin, cout, k, pad, hw):
    """qiddm_qconv_forward (unfold fused, inference) == unfold + QNode path == oracle."""
    from qiddm_amd import nn
    torch.manual_seed(31)
    m = nn.QConv2d(cin, cout, k, pad, 2).to(DEV)
    x = _img(2, hw, 3, c=cin)
    with torch.no_grad():
        fused = m(x.to(DEV)).cpu()
    traced = m(x.to(DEV)).detach().cpu()          # grad enabled -> unfold + traced QNode
    kk = k if isinstance(k, tuple) else (k, k)
    pp = pad if isinstance(pad, tuple) else (pad, pad)
    ref = oc.qconv2d_forward(x, m.weights.detach().cpu(), cout, kk, pp)
    assert fused.shape == traced.shape == ref.shape
    assert torch.allclose(fused, ref, atol=1e-3), (fused - ref).abs().max()
    assert torch.allclose(fused, traced, atol=1e-4), (fused - traced).abs().max()


@pytest.mark.parametrize("precision,tol", [("f32", 1e-4), ("f64", 1e-9)])
@pytest.mark.parametrize("n,N,L,S,P", [(8, 1, 1, 14, 784), (8, 2, 6, 2, 784), (9, 2, 2, 2, 300), (10, 1, 3, 2, 100),
                                       (6, 2, 14, 2, 784), (6, 1, 1, 14, 784), (7, 2, 3, 2, 300), (7, 1, 1, 8, 64),
                                       (4, 1, 1, 2, 64), (4, 1, 2, 2, 64), (5, 2, 3, 2, 100), (2, 1, 1, 3, 16), (3, 2, 2, 2, 30)])
def test_dense_sample_quad_kernel(n, N, L, S, P, precision, tol):
    """qiddm_dense_sample: four wavefronts per sample, several sampling-loop bodies in one launch."""
    from qiddm_amd.circuit import Circuit, dense_sample
    g = torch.Generator().manual_seed(n * 10 + P)
    B, steps = 7, 3
    x = torch.rand(B, P, generator=g, dtype=torch.float64)
    wd = torch.randn(n, P, generator=g, dtype=torch.float64) / P ** 0.5 * 3
    bd = torch.randn(n, generator=g, dtype=torch.float64)
    wu = torch.randn(P, n, generator=g, dtype=torch.float64) * 0.3
    bu = torch.rand(P, generator=g, dtype=torch.float64)
    w = torch.randn(N, L, S, n, 3, generator=g, dtype=torch.float64) * 0.6
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=N, n_blocks=L, sel_layers=S)
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz")
    net = lambda t: oc.run_circuit(spec, t @ wd.T + bd, w) @ wu.T + bu
    dev = lambda t: t.to(DEV)
    for post in (0, 1):
        got = dense_sample(circ, dev(x), dev(wd), dev(bd), dev(w), dev(wu), dev(bu), steps, precision,
                           post_mode=post, noise_factor=0.8).cpu()
        cur, refs = x, []
        for _ in range(steps):
            cur = net(cur) if post == 0 else torch.clamp(cur - (net(cur) - 0.5) * 0.1 * 0.8, 0, 1)
            refs.append(cur)
        ref = torch.stack(refs)
        assert got.shape == ref.shape
        assert torch.allclose(got, ref, atol=tol, rtol=tol), (post, (got - ref).abs().max())


def test_dense_forward_large_batch_uses_wave_kernel():
    """Above 1024 samples qiddm_dense_forward runs the one-wave-per-sample kernel (LDS-staged weights)."""
    from qiddm_amd.circuit import Circuit, dense_forward
    g = torch.Generator().manual_seed(3)
    n, P, B = 8, 784, 2100
    x = torch.rand(B, P, generator=g, dtype=torch.float64)
    wd = torch.randn(n, P, generator=g, dtype=torch.float64) * 0.1
    wu = torch.randn(P, n, generator=g, dtype=torch.float64)
    w = torch.randn(1, 2, 2, n, 3, generator=g, dtype=torch.float64) * 0.6
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_blocks=2, sel_layers=2)
    got = dense_forward(circ, x.to(DEV), wd.to(DEV), None, w.to(DEV), wu.to(DEV), None, "f32").cpu()
    ref = oc.run_circuit(oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz"), x @ wd.T, w) @ wu.T
    assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4), (got - ref).abs().max()


@pytest.mark.parametrize("ctor,side", [
    (lambda nn: nn.QIDDM_LL_noise(784, 8, 3, 2), 28),
    (lambda nn: nn.QIDDM_LL_noise(784, 6, 14, 2), 28),      # the reference's MNIST default, src/mnist_exm.py:46
    (lambda nn: nn.QNN_noise(64, 4, 2), 8),                  # BASELINE config 1, src/mnist_noise.py:49
])
def test_diffusion_sample_uses_fused_steps(ctor, side):
    """Diffusion.sample through the fused sampler == the package's step-by-step loop == ``oracle.diffusion.sample``
    around the oracle's float64 restatement of the net."""
    from oracle import diffusion as odf
    from qiddm_amd import models, nn, noise
    for goal in ("data", "noise"):
        torch.manual_seed(14)
        net = ctor(nn)
        diff = models.Diffusion(net, noise.add_normal_noise_multiple, goal, (side, side)).to(DEV, dtype=torch.double).eval()
        x = (_img(6, side, 15) * 0.75 + 0.5).to(DEV)
        mosaic = diff.sample(first_x=x, n_iters=4)
        with torch.no_grad():
            cur, outs = x, [x]
            for _ in range(4):
                cur = diff.denoise_step(cur)
                outs.append(cur)
        st = torch.stack(outs)
        ref = st[:, :, 0].permute(0, 2, 1, 3).reshape(5 * side, 6 * side)
        assert mosaic.shape == ref.shape
        assert torch.allclose(mosaic, ref, atol=1e-4), (goal, (mosaic - ref).abs().max())
        # the oracle loop (reference src/models.py:106-147 around nn/qdense.py:267-289 / :1620-1642)
        sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}
        if "weights1" in sd:
            def onet(t):
                return oc.qiddm_ll_forward(t, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights1"],
                                           sd["linear_up.weight"], sd["linear_up.bias"])
        else:
            def onet(t):
                return oc.qnn_forward(t, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights"],
                                      sd["linear_up.weight"], sd["linear_up.bias"])
        want = odf.sample(onet, x.cpu(), 4, goal)
        assert torch.allclose(mosaic.cpu(), want, atol=2e-4), (goal, (mosaic.cpu() - want).abs().max())


@pytest.mark.parametrize("shape", [(1, 3, 5, 7), (6, 8, 28, 28), (70, 4, 14, 14), (130, 2, 3, 3)])
def test_hip_batchnorm_training_matches_torch(shape):
    """qiddm_batchnorm_train_forward / _backward vs torch.nn.BatchNorm2d (training mode, float64): output, running
    statistics, num_batches_tracked and all three gradients over two steps; batch counts on either side of the
    64-slice grid and planes smaller / larger than a workgroup."""
    from qiddm_amd.circuit import batch_norm_train
    torch.manual_seed(2)
    ref = torch.nn.BatchNorm2d(shape[1], dtype=torch.float64).cuda().train()
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
    mine = torch.nn.BatchNorm2d(shape[1], dtype=torch.float64).cuda().train()
    mine.load_state_dict(ref.state_dict())
    for step in range(2):
        x = (torch.rand(*shape, dtype=torch.float64, device="cuda") * 3 + 10.0)     # mean >> std: cancellation check
        g = torch.randn(*shape, dtype=torch.float64, device="cuda")
        xa = x.clone().requires_grad_(True)
        xb = x.clone().requires_grad_(True)
        ya = ref(xa)
        yb = batch_norm_train(mine, xb)
        (ya * g).sum().backward()
        (yb * g).sum().backward()
        assert torch.allclose(ya, yb, rtol=1e-11, atol=1e-11), (ya - yb).abs().max()
        assert torch.allclose(xa.grad, xb.grad, rtol=1e-9, atol=1e-10), (xa.grad - xb.grad).abs().max()
        assert torch.allclose(ref.weight.grad, mine.weight.grad, rtol=1e-10, atol=1e-10)
        assert torch.allclose(ref.bias.grad, mine.bias.grad, rtol=1e-10, atol=1e-10)
    for (k, a), (_, b) in zip(ref.state_dict().items(), mine.state_dict().items()):
        assert torch.allclose(a.double(), b.double(), rtol=1e-11, atol=1e-12), k
    # eval mode and float32 stay with torch
    mine.eval()
    assert torch.equal(batch_norm_train(mine, x), mine(x))


@pytest.mark.parametrize("shape", [(2, 3, 1, 1), (3, 5, 7, 6), (4, 8, 14, 14), (1, 2, 2, 9)])
def test_hip_bilinear_upsample_matches_torch(shape):
    """qiddm_upsample2x_forward / _backward vs torch.nn.Upsample(scale_factor=2, mode="bilinear") in float64."""
    from qiddm_amd.nn.utils import bilinear_upsample2x
    torch.manual_seed(4)
    x = torch.randn(*shape, dtype=torch.float64, device="cuda")
    g = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], dtype=torch.float64, device="cuda")
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    ya = torch.nn.Upsample(scale_factor=2, mode="bilinear")(xa)
    yb = bilinear_upsample2x(xb)
    (ya * g).sum().backward()
    (yb * g).sum().backward()
    assert torch.allclose(ya, yb, rtol=1e-13, atol=1e-13), (ya - yb).abs().max()
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-12, atol=1e-12), (xa.grad - xb.grad).abs().max()


def test_probability_nets_fused_post_processing_tracks_weight_updates():
    """differN_noise / QDenseUndirected_old_noise at inference: circuit + `_post_process` in one launch on a gate table
    cached per weights -- equal to the oracle before and after an in-place weight update (the cache key is the version)."""
    from qiddm_amd import nn
    torch.manual_seed(5)
    net = nn.differN_noise(8, 3, 2).to(DEV).eval()
    red = torch.randn(9, 6, device=DEV)
    with torch.no_grad():
        for _ in range(2):
            got = net.forward_from_reduced(red).cpu()
            ref = oc.differn_from_reduced(red.cpu(), net.weights.detach().cpu(), (8, 8))
            assert torch.allclose(got, ref, atol=2e-3), (got - ref).abs().max()
            net.weights.mul_(1.3)
    net2 = nn.QDenseUndirected_old_noise(4, 8).to(DEV).eval()
    x = torch.rand(5, 1, 8, 8, dtype=torch.float64, device=DEV)
    with torch.no_grad():
        for _ in range(2):
            got = net2(x).cpu()
            ref = oc.qdense_undirected_forward(x.cpu(), net2.weights.detach().cpu().double(), (8, 8), weight_map="tanh")
            assert torch.allclose(got, ref, atol=2e-3), (got - ref).abs().max()
            net2.weights.add_(0.2)


@pytest.mark.parametrize("b,c,h,w,bias", [(3, 8, 28, 28, True), (1, 1, 1, 1, True), (5, 13, 7, 9, False), (2, 32, 6, 5, True),
                                          (700, 8, 4, 4, True)])
def test_hip_one_channel_head_matches_torch_conv(b, c, h, w, bias):
    """The UNets' final 1x1 convolution to one channel (qiddm_conv1x1_forward + the one-pass
    qiddm_conv1x1_head_backward) against torch's own float64 Conv2d: output and all three gradients; sizes on either
    side of the 8 / 16 / 32 channel variants and more pixels than one round of workgroups."""
    from qiddm_amd.nn.utils import pointwise_conv
    torch.manual_seed(b + c)
    ref = torch.nn.Conv2d(c, 1, 1, bias=bias).to(DEV, torch.double)
    mine = torch.nn.Conv2d(c, 1, 1, bias=bias).to(DEV, torch.double)
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(b, c, h, w, dtype=torch.float64, device=DEV)
    g = torch.randn(b, 1, h, w, dtype=torch.float64, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = (xa * ref.weight.view(1, -1, 1, 1)).sum(dim=1, keepdim=True) + (ref.bias.view(1, 1, 1, 1) if bias else 0.0)
    yb = pointwise_conv(mine, xb)
    assert type(yb.grad_fn).__name__ == "_Conv1x1HeadFunctionBackward"
    (ya * g).sum().backward()
    (yb * g).sum().backward()
    assert torch.allclose(ya, yb, rtol=1e-13, atol=1e-13)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-13, atol=1e-13)
    assert torch.allclose(ref.weight.grad, mine.weight.grad, rtol=1e-11, atol=1e-11), (ref.weight.grad - mine.weight.grad).abs().max()
    if bias:
        assert torch.allclose(ref.bias.grad, mine.bias.grad, rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("shape", [(3, 8, 28, 28), (2, 3, 7, 9), (1, 1, 2, 2), (5, 2, 3, 2), (40, 16, 14, 14)])
def test_hip_maxpool2_matches_torch(shape):
    """qiddm_maxpool2_forward / _backward against torch.nn.MaxPool2d(2, 2) in float64: odd extents (the last row / column
    belongs to no window and gets a zero gradient) and ties (the first maximum of a window takes the gradient)."""
    from qiddm_amd.circuit import max_pool2
    torch.manual_seed(sum(shape))
    pool = torch.nn.MaxPool2d(kernel_size=2, stride=2)
    x = torch.randn(*shape, dtype=torch.float64, device=DEV)
    x = torch.round(x * 2) / 2                                   # many exact ties inside the windows
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = pool(xa), max_pool2(pool, xb)
    assert type(yb.grad_fn).__name__ == "_MaxPool2FunctionBackward"
    g = torch.randn_like(ya)
    (ya * g).sum().backward()
    (yb * g).sum().backward()
    assert torch.equal(ya, yb)
    assert torch.equal(xa.grad, xb.grad)
    # other pooling geometries stay with torch
    other = torch.nn.MaxPool2d(kernel_size=3, stride=2)
    assert torch.equal(max_pool2(other, x), other(x)) if shape[2] >= 3 and shape[3] >= 3 else True


@pytest.mark.parametrize("cls_name,qdepth,side", [("QDenseUndirected_old", 3, 8), ("QDenseUndirected_old_noise", 5, 28),
                                                  ("QDenseUndirected_old_noise", 2, 5)])
def test_dense_unitary_route_matches_simulation_and_oracle(cls_name, qdepth, side, monkeypatch):
    """A4 nets at inference from 256 samples on: one float32 product with the cached circuit unitary
    (qiddm_amp_embed_rows -> GEMM -> qiddm_prob_post) against the per-sample simulation kernel and the oracle; the cache
    follows in-place weight updates; smaller batches keep the simulation."""
    from qiddm_amd import nn
    from qiddm_amd.nn import qdense
    torch.manual_seed(13)
    net = getattr(nn, cls_name)(qdepth, side).to(DEV).eval()
    x = torch.rand(300, 1, side, side, dtype=torch.float64, device=DEV)
    calls = {"unitary": 0}
    real = qdense._c.dense_unitary_forward

    def spy(*a, **k):
        calls["unitary"] += 1
        return real(*a, **k)
    monkeypatch.setattr(qdense._c, "dense_unitary_forward", spy)
    with torch.no_grad():
        for _ in range(2):
            got = net(x)
            assert calls["unitary"] >= 1
            monkeypatch.setattr(qdense, "_DENSE_UNITARY", False)
            sim = net(x)
            monkeypatch.setattr(qdense, "_DENSE_UNITARY", True)
            assert torch.allclose(got, sim, atol=2e-4), (got - sim).abs().max()
            wm = "qw_tanh" if cls_name == "QDenseUndirected_old" else "tanh"
            ref = oc.qdense_undirected_forward(x[:16].cpu(), net.weights.detach().cpu(), (side, side), wm)
            assert torch.allclose(got[:16].cpu(), ref, atol=2e-4), (got[:16].cpu() - ref).abs().max()
            net.weights.mul_(1.2)                     # the operand is rebuilt for the new weights
        before = calls["unitary"]
        small = x[:qdense._UNITARY_ROUTE_MIN_BATCH - 1]
        net(small)
        assert calls["unitary"] == before             # stale cache, below the batch threshold: the simulation kernel
        net(x)
        net(small)
        assert calls["unitary"] == before + 2         # once the unitary of these weights exists every batch takes it
