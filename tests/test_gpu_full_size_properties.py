"""BASELINE.json's full sizes (C2..C5), where the CPU oracle would take minutes: size-independent
properties of the domain instead of element-wise comparison -- norm preservation (sum of probabilities
= 1, |<Z>| <= 1), known answers (zero weights -> |0..0>; RY(pi) wire order), finding F2 (a single RZ
encoding layer on |0..0> is a global phase: the output cannot depend on the input), batch-order
independence and determinism."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _circ(**kw):
    from qiddm_amd.circuit import Circuit
    return Circuit(**kw)


def _fwd(circ, x, w, prec="f32"):
    from qiddm_amd.circuit import run_forward
    return run_forward(circ, x, w, prec)


def test_c2_full_batch_qnn_noise():
    """C2: (256, 1, 28, 28), QNN_noise(784, 8, 14)."""
    from qiddm_amd import models, nn, noise
    torch.manual_seed(42)
    net = nn.QNN_noise(784, 8, 14)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (28, 28)).to(DEV, dtype=torch.double).eval()
    x = (torch.rand(256, 1, 28, 28, dtype=torch.double) * 0.75 + 0.5).to(DEV)
    with torch.no_grad():
        y1 = diff.denoise_step(x)
        y2 = diff.denoise_step(torch.rand_like(x))           # F2: output independent of the input
        steps = diff.denoise_steps(x, 15)                     # tau_test = 15 (src/mnist_exm.py:211)
    assert torch.allclose(y1, y2, atol=1e-4)
    assert steps.shape == (15, 256, 1, 28, 28) and torch.isfinite(steps).all()
    assert torch.allclose(steps[0], y1, atol=1e-6) and torch.allclose(steps[14], y1, atol=1e-4)
    perm = torch.randperm(256, device=DEV)
    with torch.no_grad():
        assert torch.equal(diff.denoise_step(x[perm]), diff.denoise_step(x)[perm])   # sample independence
        assert torch.equal(diff.denoise_step(x), y1)                                 # determinism


def test_c3_full_batch_10_qubits():
    """C3: batch 1024, n = 10, differN_noise(28, 9, 2) circuit and the 60-layer CNOT circuit."""
    torch.manual_seed(1)
    circ = _circ(n_qubits=10, encoding="rz", imprimitive="CZ", measure="probs", n_rounds=2, n_blocks=9, sel_layers=2)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
    x = torch.randn(1024, 10, device=DEV)
    p = _fwd(circ, x, w)
    assert p.shape == (1024, 1024) and (p >= 0).all()
    assert torch.allclose(p.sum(1), torch.ones(1024, device=DEV), atol=2e-5)
    e0 = torch.zeros(1024, 1024, device=DEV)
    e0[:, 0] = 1
    assert torch.allclose(_fwd(circ, x, torch.zeros_like(w)), e0, atol=1e-6)          # KA1
    deep = _circ(n_qubits=10, encoding="amplitude", imprimitive="CNOT", measure="probs", sel_layers=60,
                 n_features=784, pad_with=0.1)
    wd = (torch.tanh(torch.randn(deep.angles_shape, dtype=torch.float64) * 0.4)).to(DEV)
    img = torch.rand(1024, 784, device=DEV)
    pd = _fwd(deep, img, wd)
    assert torch.allclose(pd.sum(1), torch.ones(1024, device=DEV), atol=5e-5)        # KA4 after 1201 gates
    # zero-angle CNOT rings only permute: the multiset of probabilities is the embedded one
    p0 = _fwd(deep, img, torch.zeros_like(wd))
    v = torch.cat([img, torch.full((1024, 240), 0.1, device=DEV)], 1)
    ref = (v / v.norm(dim=1, keepdim=True)) ** 2
    assert torch.allclose(p0.sort(dim=1).values, ref.sort(dim=1).values, atol=1e-6)


def test_c4_12_qubit_qconv_circuit():
    """C4: 12-qubit QConv2d circuit (C_in = 256, k = 3 -> 2304 features); 32 images' worth of pixels."""
    torch.manual_seed(2)
    circ = _circ(n_qubits=12, encoding="amplitude", imprimitive="CNOT", measure="probs", sel_layers=3,
                 n_features=2304, pad_with=0.5, enc_offset=0.1)
    w = (math.pi * torch.tanh(torch.rand(circ.angles_shape, dtype=torch.float64) * math.pi - math.pi / 2)).to(DEV)
    feats = torch.rand(32 * 1024, 2304, device=DEV)
    p = _fwd(circ, feats, w)
    assert p.shape == (32 * 1024, 4096)
    assert torch.allclose(p.sum(1), torch.ones(p.shape[0], device=DEV), atol=5e-5)
    again = _fwd(circ, feats[:1000], w)
    assert torch.equal(again, p[:1000])


def test_c5_16_qubits_batch_1024():
    """C5: 16 qubits (65 536 amplitudes), 1024 samples per GPU, LL-style (.., 16, 6, 2) circuit."""
    torch.manual_seed(3)
    circ = _circ(n_qubits=16, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=2, n_blocks=6, sel_layers=2)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
    x = torch.randn(1024, 16, device=DEV)
    ev = _fwd(circ, x, w)
    assert ev.shape == (1024, 16) and torch.isfinite(ev).all() and ev.abs().max() <= 1 + 1e-5
    assert torch.allclose(_fwd(circ, x, torch.zeros_like(w)), torch.ones(1024, 16, device=DEV), atol=1e-6)   # KA1
    # F2 at n = 16: one encoding layer in front of the ansatz cannot influence the output
    one = _circ(n_qubits=16, encoding="rz", imprimitive="CZ", measure="expz", sel_layers=3)
    w1 = (torch.randn(one.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
    a, b = _fwd(one, x[:64], w1), _fwd(one, torch.zeros(64, 16, device=DEV), w1)
    assert torch.allclose(a, b, atol=1e-5)
    # KA3 at n = 16: RY(pi) on wire 0 only -> <Z_0> = -1, every other wire +1
    w0 = torch.zeros(one.angles_shape, dtype=torch.float64)
    w0[0, 0, 0, 0, 1] = math.pi
    ev0 = _fwd(_circ(n_qubits=16, encoding="rz", imprimitive="CZ", measure="expz", sel_layers=1),
               x[:4], w0[:, :, :1].contiguous().to(DEV))
    expect = torch.ones(4, 16, device=DEV)
    expect[:, 0] = -1
    assert torch.allclose(ev0, expect, atol=1e-6)
    probs = _circ(n_qubits=16, encoding="rz", imprimitive="CNOT", measure="probs", n_blocks=2, sel_layers=2)
    wp = (torch.randn(probs.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
    p = _fwd(probs, x[:256], wp)
    assert torch.allclose(p.sum(1), torch.ones(256, device=DEV), atol=5e-5)


@pytest.mark.parametrize("c_in,c_out,k,pad", [(16, 8, 3, 1), (32, 16, 3, 1), (16, 8, 1, 0)])
def test_qconv_training_routes_agree_at_full_resolution(c_in, c_out, k, pad):
    """A unet_simple layer at 28 x 28, 64 net samples (50 176 circuits -- beyond what the oracle finishes in seconds):
    the float32 unitary route (GEMM forward, thin-product backward, one adjoint sweep per channel) against the
    float64 per-pixel route (fused circuit launch, adjoint sweep per output pixel, fold) -- two independent
    algorithms for the same layer -- on outputs, weight gradients and input gradients."""
    from qiddm_amd import nn, set_default_precision
    torch.manual_seed(21)
    layer = nn.QConv2d(c_in, c_out, k, pad, 3).cuda().train()
    x = torch.rand(64, c_in, 28, 28, dtype=torch.float64, device="cuda")
    g = torch.randn(64, c_out, 28, 28, dtype=torch.float64, device="cuda")
    res = {}
    for precision in ("f32", "f64"):
        set_default_precision(precision)
        try:
            layer.weights.grad = None
            xi = x.clone().requires_grad_(True)
            y = layer(xi)
            (y * g).sum().backward()
            res[precision] = (y.detach(), layer.weights.grad.clone(), xi.grad.clone())
        finally:
            set_default_precision("f32")
    (y32, gw32, gx32), (y64, gw64, gx64) = res["f32"], res["f64"]
    assert (y32 - y64).abs().max().item() < 2e-4
    assert (gw32 - gw64).abs().max().item() < 1e-3 * max(gw64.abs().max().item(), 1.0)
    assert (gx32 - gx64).abs().max().item() < 1e-3 * max(gx64.abs().max().item(), 1.0)
    assert 0.0 <= y32.min().item() and y32.max().item() <= 1.0


def test_c3_unet_simple_at_batch_1024():
    """C3's net at its full batch: ``UNetUndirectedS(3, 8, 3)`` on (1024, 1, 28, 28) through the eval-mode route
    (circuit unitaries + MFMA GEMMs).  The oracle takes ~0.6 s per image, so: 8 images of the batch against the
    oracle, and batch-composition independence (eval mode has no cross-sample term) for all 1024."""
    from oracle import unet as ou
    from qiddm_amd import nn
    torch.manual_seed(31)
    net = nn.UNetUndirectedS(3, 8, 3).to(DEV, dtype=torch.double).eval()
    x = torch.rand(1024, 1, 28, 28, dtype=torch.double, device=DEV) * 0.75 + 0.5
    with torch.no_grad():
        y = net(x)
        assert y.shape == (1024, 1, 28, 28) and torch.isfinite(y).all()
        parts = torch.cat([net(x[i:i + 256]) for i in range(0, 1024, 256)])
        assert torch.allclose(y, parts, atol=1e-9), (y - parts).abs().max()
        perm = torch.randperm(1024, device=DEV)
        assert torch.allclose(net(x[perm]), y[perm], atol=1e-9)
    idx = torch.tensor([0, 1, 255, 256, 511, 700, 1022, 1023])
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = ou.unet_simple_forward(x[idx.to(DEV)].cpu(), sd, 3, 8, training=False)
    assert torch.allclose(y[idx.to(DEV)].cpu(), want, atol=5e-4, rtol=5e-4), (y[idx.to(DEV)].cpu() - want).abs().max()


def test_c4_qconv_layer_at_the_512_image_shard():
    """C4's layer at the per-GPU shard: 12-qubit ``QConv2d(256, 256, 3x3, qdepth 3)`` on (512, 256, 32, 32) float64
    (1 GiB in, 1 GiB out; 524 288 circuits) through the eval GEMM route.  Checked against the statevector oracle on
    48 output pixels spread over the shard (corners, edges, interior), plus range, determinism and chunk equality."""
    from oracle import circuits as oc
    from qiddm_amd import nn
    torch.manual_seed(33)
    conv = nn.QConv2d(256, 256, qdepth=3).to(DEV).eval()
    x = torch.rand(512, 256, 32, 32, dtype=torch.double, device=DEV)
    with torch.no_grad():
        y = conv(x)
        assert y.shape == (512, 256, 32, 32)
        assert 0.0 <= y.min().item() and y.max().item() <= 1.0
        assert torch.equal(conv(x[100:132]), y[100:132])              # chunk == slice of the shard, bit for bit
    g = torch.Generator().manual_seed(34)
    pix = [(0, 0, 0), (511, 31, 31), (17, 0, 31), (300, 31, 0), (256, 15, 16)]
    pix += [(int(torch.randint(512, (1,), generator=g)), int(torch.randint(32, (1,), generator=g)),
             int(torch.randint(32, (1,), generator=g))) for _ in range(43)]
    w = conv.weights.detach().cpu()
    for b, i, j in pix:
        # the 3x3 patch around (i, j) with zero padding, as a 1-pixel "image": the oracle's unfold of a 3x3 input
        # with padding 0 yields exactly that patch
        patch = torch.zeros(1, 256, 3, 3, dtype=torch.double)
        i0, i1, j0, j1 = max(i - 1, 0), min(i + 2, 32), max(j - 1, 0), min(j + 2, 32)
        patch[0, :, i0 - (i - 1):i1 - (i - 1), j0 - (j - 1):j1 - (j - 1)] = x[b, :, i0:i1, j0:j1].cpu()
        want = oc.qconv2d_forward(patch, w, 256, (3, 3), (0, 0))[0, :, 0, 0]
        got = y[b, :, i, j].cpu()
        assert torch.allclose(got, want, atol=2e-4), ((b, i, j), (got - want).abs().max())
