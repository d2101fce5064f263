"""Eval-mode quantum convolution: circuit unitary + MFMA GEMM (reference
nn/qconv.py:92-126) against the oracle's statevector route and the package's circuit-simulation kernel."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n,imp", [(1, "CNOT"), (2, "CNOT"), (3, "CZ"), (5, "CNOT"), (6, "CNOT"), (8, "CNOT"),
                                   (10, "CNOT"), (7, "CZ")])
def test_circuit_unitary_vs_oracle(n, imp):
    from oracle import statevector as sv
    from qiddm_amd import circuit as qc
    torch.manual_seed(n)
    w = torch.randn(3, n, 3, dtype=torch.float64) * 0.8
    d = 1 << n
    # oracle: apply the layers to the identity's columns
    cols = sv.strongly_entangling_layers(torch.eye(d, dtype=torch.complex128).reshape((d,) + (2,) * n), w, n, imp)
    want = cols.reshape(d, d).T                       # want[k, j] = <k|U|j>
    got = qc.circuit_unitary(w.to(DEV), n, imp, precision="f64").cpu()
    assert (got - want).abs().max().item() < 1e-12
    assert (got @ got.conj().T - torch.eye(d, dtype=torch.complex128)).abs().max().item() < 1e-12
    got32 = qc.circuit_unitary(w.to(DEV), n, imp, precision="f32").cpu()
    assert (got32 - want).abs().max().item() < 5e-6


CASES = [  # (C_in, C_out, k, pad, H, W, B, qdepth)
    (1, 8, 3, 1, 28, 28, 3, 3),       # unet_simple first layer: n = 4
    (8, 8, 3, 1, 14, 14, 2, 3),       # n = 7
    (16, 16, 3, 1, 7, 7, 2, 2),       # n = 8
    (3, 5, 3, 1, 9, 11, 2, 2),        # n = 5, ragged sizes
    (32, 32, 3, 1, 7, 7, 1, 2),       # n = 9, K = 288
    (8, 40, 3, 0, 8, 8, 1, 1),        # n = 7, two channel tiles, no padding
    (1, 1, 1, 0, 4, 4, 1, 1),         # n = 1: the single even-index probability
    (48, 64, 3, 1, 6, 6, 1, 1),       # n = 9, K = 432, two channel tiles
    (48, 200, 3, 1, 6, 5, 2, 1),      # n = 9, 200 channels: the 128-channels-per-workgroup kernel, ragged last group
]


@pytest.mark.parametrize("case", CASES)
def test_qconv_unitary_vs_oracle_and_circuit_kernel(case):
    from oracle import circuits as oc
    from qiddm_amd import nn
    cin, cout, k, pad, h, w, b, qd = case
    torch.manual_seed(sum(case))
    layer = nn.QConv2d(cin, cout, kernel_size=k, padding=pad, qdepth=qd).to(DEV)
    x = torch.rand(b, cin, h, w, dtype=torch.float64, device=DEV)
    want = oc.qconv2d_forward(x.cpu(), layer.weights.detach().cpu(), cout, (k, k), (pad, pad))
    with torch.no_grad():
        layer.train()
        sim = layer(x).cpu()                      # circuit simulation, one wavefront per pixel
        layer.eval()
        assert layer.sample_matrix is None
        got = layer(x).cpu()                      # unitary + GEMM
        assert layer.sample_matrix is not None and layer.sample_matrix.shape == (2 ** layer.wires,) * 2
    assert got.shape == want.shape
    # outputs are probabilities * D/2 clamped to [0, 1]: absolute tolerance scales with D/2 (float32 products)
    tol = 2e-5 * max(1.0, 2 ** layer.wires / 2) / 8
    assert (got - want).abs().max().item() < tol
    assert (sim - want).abs().max().item() < tol
    layer.train()
    assert layer.sample_matrix is None            # reference :123-125


def test_unitary_cache_follows_the_weights():
    from qiddm_amd import nn
    torch.manual_seed(0)
    layer = nn.QConv2d(2, 4, qdepth=1).to(DEV).eval()
    x = torch.rand(1, 2, 5, 5, dtype=torch.float64, device=DEV)
    with torch.no_grad():
        y0 = layer(x)
        u0 = layer.sample_matrix
        assert layer(x) is not None and layer.sample_matrix is u0          # reused
        layer.weights.add_(0.3)
        y1 = layer(x)
    assert layer.sample_matrix is not u0 and not torch.allclose(y0, y1)


def test_unet_simple_eval_matches_train_mode_inference():
    """The whole unet_simple net: eval-mode (GEMM) convolutions == circuit-simulation convolutions, with the
    BatchNorm layers held in eval mode for both."""
    from qiddm_amd import nn
    torch.manual_seed(1)
    net = nn.UNetUndirectedS(2, 4, 2).to(DEV).to(torch.double).eval()
    x = torch.rand(2, 1, 16, 16, dtype=torch.float64, device=DEV)
    with torch.no_grad():
        y_gemm = net(x)
        for m in net.modules():
            if isinstance(m, nn.QConv2d):
                m.training = True                      # circuit route for the convolutions only
        y_sim = net(x)
    assert torch.allclose(y_gemm, y_sim, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("hs,ws", [(7, 7), (5, 8), (1, 3), (14, 14)])
def test_fused_upsample_matches_torch_upsample_then_conv(hs, ws):
    from qiddm_amd import nn
    torch.manual_seed(hs * 31 + ws)
    layer = nn.QConv2d(6, 3, kernel_size=1, padding=0, qdepth=2).to(DEV).eval()
    x = torch.rand(3, 6, hs, ws, dtype=torch.float64, device=DEV)
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear")
    with torch.no_grad():
        want = layer(up(x))
        got = layer.eval_forward(x, upsample2x=True)
    assert got.shape == want.shape == (3, 3, 2 * hs, 2 * ws)
    assert torch.allclose(got, want, atol=2e-6, rtol=0)


def test_fused_batchnorm_epilogue_and_conv1x1():
    from qiddm_amd import circuit as qc
    from qiddm_amd import nn
    torch.manual_seed(3)
    layer = nn.QConv2d(4, 8, qdepth=2).to(DEV).eval()
    bn = torch.nn.BatchNorm2d(8, dtype=torch.float64).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(0.0, 0.5)
        bn.running_var.uniform_(0.2, 2.0)
    bn.eval()
    x = torch.rand(2, 4, 9, 9, dtype=torch.float64, device=DEV)
    with torch.no_grad():
        want = bn(layer(x))
        got = layer.eval_forward(x, batch_norm=bn)
        assert torch.allclose(got, want, atol=1e-12, rtol=1e-12)      # same GEMM, affine epilogue in float64
        bn.train()
        assert layer.eval_forward(x, batch_norm=bn) is None           # batch statistics: not folded
        conv = torch.nn.Conv2d(8, 3, kernel_size=1).to(DEV).double()
        assert torch.allclose(qc.conv1x1_forward(want, conv.weight, conv.bias), conv(want), atol=1e-13, rtol=1e-13)
        conv_nb = torch.nn.Conv2d(8, 1, kernel_size=1, bias=False).to(DEV).double()
        assert torch.allclose(qc.conv1x1_forward(want, conv_nb.weight, None), conv_nb(want), atol=1e-13, rtol=1e-13)


def test_unet_simple_fused_inference_vs_module_by_module():
    """UNetUndirectedS in eval mode (fused blocks) == the same modules run one by one through torch's
    Upsample / BatchNorm2d / Conv2d around the unfused eval-mode convolutions."""
    from qiddm_amd import nn
    from qiddm_amd.nn.unet import UNetUndirected
    torch.manual_seed(5)
    net = nn.UNetUndirectedS(3, 8, 2).to(DEV).to(torch.double)
    with torch.no_grad():                               # non-trivial running statistics
        net.train()
        for _ in range(2):
            net(torch.rand(4, 1, 28, 28, dtype=torch.float64, device=DEV))
    net.eval()
    x = torch.rand(3, 1, 28, 28, dtype=torch.float64, device=DEV)
    with torch.no_grad():
        got = net(x)
        # reference wiring, module by module: the base classes' forwards never take the fused shortcuts
        xx, skips = x, []
        for blk in net.down_blocks:
            before = blk.net(xx)
            skips.append(before)
            xx = blk.pooling_layer(before) if blk.pooling else before
        for i, blk in enumerate(net.up_blocks):
            from qiddm_amd.nn.utils import autopad
            skip, up = autopad(skips[-(i + 2)], blk.up_conv(xx))
            xx = blk.net(torch.cat([up, skip], dim=1))
        want = net.final_conv(xx)
    assert torch.allclose(got, want, atol=5e-6, rtol=1e-6)


def test_float64_setting_keeps_the_float64_circuit_kernel():
    from oracle import circuits as oc
    from qiddm_amd import circuit as qc
    from qiddm_amd import nn
    torch.manual_seed(9)
    layer = nn.QConv2d(2, 4, qdepth=2).to(DEV).eval()
    x = torch.rand(1, 2, 5, 5, dtype=torch.float64, device=DEV)
    want = oc.qconv2d_forward(x.cpu(), layer.weights.detach().cpu(), 4, (3, 3), (1, 1))
    prev = qc._default_precision
    qc.set_default_precision("f64")
    try:
        with torch.no_grad():
            got = layer(x).cpu()
        assert layer.eval_forward(x) is None
    finally:
        qc.set_default_precision(prev)
    assert (got - want).abs().max().item() < 1e-11


def test_wide_unitary_and_twelve_qubit_eval_convolution():
    """n = 11 circuit unitary (written transposed, returned as a view) against the oracle, and BASELINE config 4's
    12-wire layer shape (C_in = 256, 3x3, 64 output channels) through the eval-mode GEMM route."""
    from oracle import circuits as oc
    from oracle import statevector as sv
    from qiddm_amd import circuit as qc
    from qiddm_amd import nn
    torch.manual_seed(11)
    n = 11
    w = torch.randn(2, n, 3, dtype=torch.float64) * 0.8
    d = 1 << n
    cols = sv.strongly_entangling_layers(torch.eye(d, dtype=torch.complex128).reshape((d,) + (2,) * n), w, n, "CNOT")
    want = cols.reshape(d, d).T
    got = qc.circuit_unitary(w.to(DEV), n, "CNOT", precision="f64")
    assert got.shape == (d, d) and not got.is_contiguous()
    assert (got.cpu() - want).abs().max().item() < 1e-12
    layer = nn.QConv2d(256, 64, qdepth=2).to(DEV).eval()
    assert layer.wires == 12
    x = torch.rand(1, 256, 4, 4, dtype=torch.float64, device=DEV)
    ref = oc.qconv2d_forward(x.cpu(), layer.weights.detach().cpu(), 64, (3, 3), (1, 1))
    with torch.no_grad():
        y = layer(x)
        layer.train()
        y_sim = layer(x)                       # tiled circuit simulation, one workgroup per pixel
    assert layer.sample_matrix is None
    tol = 2e-5 * (2 ** 12 / 2) / 8
    assert (y.cpu() - ref).abs().max().item() < tol
    assert (y_sim.cpu() - ref).abs().max().item() < tol


def test_packed_operand_is_kept_per_layer_and_follows_weights_and_batchnorm(monkeypatch):
    """Eval-mode layers pack the GEMM operand once per (weights, BatchNorm) state: the second call hands the library
    u = NULL (no pack launch) and returns the same numbers; an in-place change of the weights or of the BatchNorm's running
    statistics packs again."""
    from qiddm_amd import _capi, nn
    torch.manual_seed(1)
    layer = nn.QConv2d(8, 8, 3, 1, 2).to(DEV).eval()
    bn = torch.nn.BatchNorm2d(8).to(DEV, torch.double).eval()
    with torch.no_grad():
        bn.running_mean.uniform_(0.1, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
    x = torch.rand(3, 8, 14, 14, dtype=torch.float64, device=DEV)
    lib = _capi.lib()
    real = lib.qiddm_qconv_unitary_forward
    packs = []

    class Spy:
        def __call__(self, n_qubits, u_ptr, *rest):
            packs.append(u_ptr != 0)
            return real(n_qubits, u_ptr, *rest)
    monkeypatch.setattr(lib, "qiddm_qconv_unitary_forward", Spy())
    with torch.no_grad():
        y0 = layer.eval_forward(x, batch_norm=bn)
        y1 = layer.eval_forward(x, batch_norm=bn)
        assert packs == [True, False] and torch.equal(y0, y1)
        bn.running_mean.add_(0.05)                         # the folded BatchNorm is part of the packing
        y2 = layer.eval_forward(x, batch_norm=bn)
        assert packs[-1] is True and not torch.equal(y2, y1)
        ref = torch.nn.functional.batch_norm(layer(x), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        assert torch.allclose(y2, ref, atol=1e-9)
        layer.weights.mul_(1.1)
        n = len(packs)
        y3 = layer.eval_forward(x, batch_norm=bn)
        assert packs[n] is True and not torch.allclose(y3, y2)
        y4 = layer.eval_forward(x, batch_norm=bn)
        assert packs[-1] is False and torch.equal(y3, y4)
