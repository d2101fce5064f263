"""GPU parity of the adjoint (reverse-mode) backward against torch autograd through the CPU oracle --
what the reference's diff_method="backprop" QNodes compute (nn/qdense.py:37, 419; nn/qconv.py:46)."""
import pytest
import torch

from oracle import circuits as oc

pytestmark = pytest.mark.gpu


def _case(n, enc, imp, meas, L, S, batch, seed, feat=None, pad=0.0, offset=0.0, scale=1.0):
    from qiddm_amd.circuit import Circuit
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(1, L, S, n, 3, generator=g, dtype=torch.float64) * 0.8
    f = feat if feat is not None else n
    x = torch.rand(batch, f, generator=g, dtype=torch.float64) * 2 - 0.5
    if enc == "amplitude":
        x = x.abs() + 0.05
    circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=meas, n_blocks=L, sel_layers=S,
                   n_features=f if enc == "amplitude" else 0, pad_with=pad, enc_offset=offset, enc_scale=scale)
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure=meas, pad_with=pad, enc_offset=offset,
                   enc_scale=scale)
    cols = 2 ** n if meas == "probs" else n
    gout = torch.randn(batch, cols, generator=g, dtype=torch.float64)
    return circ, spec, x, w, gout


def _oracle_grads(spec, x, w, gout):
    w = w.clone().requires_grad_(True)
    x = x.clone().requires_grad_(True)
    loss = (oc.run_circuit(spec, x, w) * gout).sum()
    return torch.autograd.grad(loss, [w, x])


CASES = [
    (1, "rz", "CZ", "expz", 2, 2), (2, "rz", "CNOT", "probs", 2, 3), (3, "rz", "CZ", "expz", 2, 2),
    (4, "rz", "CZ", "probs", 3, 2), (5, "ry", "CNOT", "probs", 1, 3), (6, "rz", "CNOT", "expz", 2, 3),
    (7, "ry", "CNOT", "probs", 1, 3), (8, "rz", "CZ", "expz", 2, 2), (8, "rz", "CNOT", "probs", 2, 9),
    (9, "rz", "CZ", "probs", 1, 2), (10, "rz", "CZ", "expz", 2, 2), (10, "rz", "CNOT", "probs", 1, 3),
    (4, "amplitude", "CNOT", "probs", 1, 3), (7, "amplitude", "CNOT", "probs", 1, 2),
    (10, "amplitude", "CNOT", "probs", 1, 2), (3, "amplitude", "CZ", "expz", 1, 2),
]


@pytest.mark.parametrize("precision,tol", [("f64", dict(atol=1e-9, rtol=1e-9)), ("f32", dict(atol=3e-4, rtol=3e-3))])
@pytest.mark.parametrize("n,enc,imp,meas,L,S", CASES)
def test_adjoint_matches_autograd(n, enc, imp, meas, L, S, precision, tol):
    from qiddm_amd.circuit import run_adjoint
    feat = {4: 9, 7: 100, 10: 784, 3: 8}.get(n) if enc == "amplitude" else None
    circ, spec, x, w, gout = _case(n, enc, imp, meas, L, S, batch=21, seed=n * 13 + L, feat=feat, pad=0.3,
                                   offset=0.1 if enc == "amplitude" else 0.0, scale=1.3 if enc != "amplitude" else 1.0)
    ga, gi = run_adjoint(circ, x.cuda(), w.cuda(), gout.cuda(), precision)
    torch.cuda.synchronize()
    ra, ri = _oracle_grads(spec, x, w, gout)
    assert torch.allclose(ga.cpu(), ra, **tol), (ga.cpu() - ra).abs().max()
    cols = gi.shape[1]
    assert torch.allclose(gi.cpu(), ri[:, :cols], **tol), (gi.cpu() - ri[:, :cols]).abs().max()


def test_adjoint_equals_parameter_shift():
    from qiddm_amd.circuit import run_adjoint, run_shift_sweep
    circ, spec, x, w, gout = _case(8, "rz", "CZ", "expz", 3, 2, batch=300, seed=77)
    a1, i1 = run_adjoint(circ, x.cuda(), w.cuda(), gout.cuda(), "f64")
    a2, i2 = run_shift_sweep(circ, x.cuda(), w.cuda(), gout.cuda(), "f64")
    assert torch.allclose(a1, a2, atol=1e-9) and torch.allclose(i1, i2, atol=1e-10)


def test_qconv_stack_trains_through_the_adjoint():
    """Two stacked QConv2d layers: the inner layer's features need d/d(amplitude-embedded input)."""
    from qiddm_amd import nn, set_default_precision
    torch.manual_seed(5)
    c1 = nn.QConv2d(1, 4, 3, 1, 2).cuda()
    c2 = nn.QConv2d(4, 2, 3, 1, 2).cuda()
    x = torch.rand(2, 1, 5, 5, dtype=torch.float64)
    set_default_precision("f64")
    try:
        y = c2(c1(x.cuda()))
        y.square().sum().backward()
    finally:
        set_default_precision("f32")
    w1 = c1.weights.detach().cpu().clone().requires_grad_(True)
    w2 = c2.weights.detach().cpu().clone().requires_grad_(True)
    ref = oc.qconv2d_forward(oc.qconv2d_forward(x, w1, 4), w2, 2)
    ref.square().sum().backward()
    assert torch.allclose(y.detach().cpu(), ref.detach(), atol=1e-10)
    assert torch.allclose(c2.weights.grad.cpu(), w2.grad, atol=1e-8), (c2.weights.grad.cpu() - w2.grad).abs().max()
    assert torch.allclose(c1.weights.grad.cpu(), w1.grad, atol=1e-8), (c1.weights.grad.cpu() - w1.grad).abs().max()


@pytest.mark.parametrize("c_in,c_out,k,pad,hw,qdepth,precision,tol", [
    (1, 4, 3, 1, (6, 5), 2, "f64", 1e-9),       # first UNet layer shape: 9 features on 4 wires
    (8, 16, 3, 1, (7, 7), 3, "f64", 1e-9),      # 72 features on 7 wires, sub-wave layout boundary
    (16, 8, 1, 0, (4, 6), 2, "f64", 1e-9),      # the 1x1 `up_conv`
    (3, 2, 3, 0, (6, 6), 1, "f64", 1e-9),       # no padding: H_out < H
    (16, 16, 3, 1, (8, 8), 3, "f32", 2e-4),     # 144 features on 8 wires, float32 engine
    (32, 32, 3, 1, (5, 5), 2, "f32", 2e-4),     # 288 features on 9 wires (per-pixel sweep: 32 channels x 289 columns)
    # float32 default: training through the circuit unitary (GEMM forward, thin-product backward, one sweep per channel)
    (1, 8, 3, 1, (6, 5), 3, "f32", 2e-4),       # first UNet layer
    (16, 8, 1, 0, (4, 6), 2, "f32", 2e-4),      # 1x1 up_conv, no pad columns to speak of (16 features on 4 wires)
    (32, 16, 3, 1, (5, 5), 2, "f32", 2e-4),     # 288 features: two column chunks per thread
    (3, 2, 3, 0, (9, 40), 1, "f32", 2e-4),      # several pixel tiles, ragged last tile, no padding
    (16, 32, 3, 1, (7, 7), 3, "f32", 2e-4),     # the third down block of unet_simple: 144 features, 32 channels
    (8, 12, 3, 1, (6, 6), 2, "f32", 2e-4),      # 12 channels in the 16-channel kernel: zero channel columns
    (4, 20, 3, 1, (5, 5), 2, "f32", 2e-4),      # 20 channels in the 32-channel kernel, 36 features
    (28, 8, 3, 1, (4, 4), 2, "f32", 2e-4),      # 252 features on 8 wires: almost no pad columns, 16 column blocks
])
def test_fused_qconv_backward_vs_oracle_autograd(c_in, c_out, k, pad, hw, qdepth, precision, tol):
    """qiddm_qconv_backward (adjoint sweep with the patch and dL/dy read in place, then the fold): d/dweights and
    d/dx of QConv2d against torch autograd through the oracle's unfold / circuit / clamp / slices."""
    from qiddm_amd import nn, set_default_precision
    torch.manual_seed(11)
    layer = nn.QConv2d(c_in, c_out, k, pad, qdepth).cuda().train()
    x = torch.rand(2, c_in, *hw, dtype=torch.float64)
    x[0, :, 0, 0] = 0.0
    gy = torch.randn(2, c_out, hw[0] + 2 * pad - k + 1, hw[1] + 2 * pad - k + 1, dtype=torch.float64)
    xg = x.cuda().requires_grad_(True)
    set_default_precision(precision)
    try:
        y = layer(xg)
        (y * gy.cuda()).sum().backward()
    finally:
        set_default_precision("f32")
    xo = x.clone().requires_grad_(True)
    wo = layer.weights.detach().cpu().clone().requires_grad_(True)
    yo = oc.qconv2d_forward(xo, wo, c_out, (k, k), (pad, pad))
    (yo * gy).sum().backward()
    assert y.shape == yo.shape
    assert torch.allclose(y.detach().cpu(), yo.detach(), atol=tol * 10)
    scale = max(1.0, wo.grad.abs().max().item())
    assert torch.allclose(layer.weights.grad.cpu(), wo.grad, atol=tol * scale * 10), \
        (layer.weights.grad.cpu() - wo.grad).abs().max()
    sx = max(1.0, xo.grad.abs().max().item())
    assert torch.allclose(xg.grad.cpu(), xo.grad, atol=tol * sx * 10), (xg.grad.cpu() - xo.grad).abs().max()


@pytest.mark.parametrize("c_in,c_out,k,pad,hw", [
    (16, 8, 3, 1, (28, 28)),     # unet_simple shapes with the kernel extent / channel count compiled in ...
    (8, 16, 3, 1, (14, 14)),
    (32, 16, 1, 0, (14, 14)),
    (32, 16, 3, 1, (14, 14)),
    (16, 32, 3, 1, (7, 7)),
    (12, 8, 3, 1, (9, 11)),      # ... and run-time shapes: 108 features, channels not a multiple of the four waves
    (6, 16, 5, 2, (8, 8)),       # 150 features, 5 x 5 taps
    (40, 8, 1, 0, (6, 6)),       # 1 x 1, 40 features
])
def test_thin_product_backward_float32_activations_and_generic_walk(c_in, c_out, k, pad, hw, monkeypatch):
    """qiddm_qconv_train_backward_x32 (float32 copy of the activations) == qiddm_qconv_train_backward bit for bit:
    every patch element is converted to float32 before the products either way.  The last three shapes have no
    compiled-in variant and take the run-time walk of the gather; all of them against autograd through the oracle."""
    from qiddm_amd import nn, circuit
    torch.manual_seed(5)
    layer = nn.QConv2d(c_in, c_out, k, pad, 2).cuda().train()
    x = torch.rand(3, c_in, *hw, dtype=torch.float64, device="cuda")
    gy = torch.randn(3, c_out, hw[0] + 2 * pad - k + 1, hw[1] + 2 * pad - k + 1, dtype=torch.float64, device="cuda")

    def grads(x32, dx=False):
        monkeypatch.setattr(circuit, "_QCONV_X32", x32)
        monkeypatch.setattr(circuit, "_QCONV_DX", dx)
        xg = x.clone().requires_grad_(True)
        gx, gw = torch.autograd.grad(layer(xg), [xg, layer.weights], gy)
        return gx, gw

    gx64, gw64 = grads(False)
    gx32, gw32 = grads(True)
    assert torch.equal(gx64, gx32) and torch.equal(gw64, gw32)
    # dL/dx from the per-pixel rows (qiddm_qconv_train_backward_dx: no feature-gradient matrix, the taps summed in float32
    # on the matrix cores) against the fold route: same weight gradients bit for bit, dL/dx to float32 rounding
    lib = circuit._capi.lib()
    co = circuit._row_channels(c_out)
    takes_dx = lib.qiddm_qconv_train_dx_elems(layer.wires, 3, c_in, hw[0], hw[1], k, k, pad, pad, c_out, co) > 0
    assert takes_dx == (c_in <= 32), "every same-size layer up to 32 input channels takes the per-pixel-row route"
    gxd, gwd = grads(False, dx=True)
    assert torch.equal(gwd, gw64)
    assert torch.allclose(gxd, gx64, rtol=0, atol=2e-6 * max(1.0, gx64.abs().max().item())), (gxd - gx64).abs().max()
    # against autograd through the oracle
    xo = x.cpu().requires_grad_(True)
    wo = layer.weights.detach().cpu().clone().requires_grad_(True)
    (oc.qconv2d_forward(xo, wo, c_out, (k, k), (pad, pad)) * gy.cpu()).sum().backward()
    assert torch.allclose(gx64.cpu(), xo.grad, atol=2e-3 * max(1.0, xo.grad.abs().max().item()))
    assert torch.allclose(gw64.cpu(), wo.grad, atol=2e-3 * max(1.0, wo.grad.abs().max().item()))


def test_differn_backprop_training_step():
    """differN (diff_method='backprop') end to end: chained rounds, grads == oracle autograd."""
    from qiddm_amd import nn, set_default_precision
    torch.manual_seed(6)
    m = nn.differN_noise(8, 3, 2).cuda()
    red = torch.randn(9, 6) * 1.5
    set_default_precision("f64")
    try:
        out = m.forward_from_reduced(red.cuda())
        out.sum().backward()
    finally:
        set_default_precision("f32")
    w = m.weights.detach().cpu().clone().requires_grad_(True)
    ref = oc.differn_from_reduced(red, w, (8, 8))
    ref.sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=1e-9)
    assert torch.allclose(m.weights.grad.cpu().double(), w.grad.double(), atol=1e-7)


@pytest.mark.parametrize("n,enc,imp,meas", [(11, "rz", "CZ", "expz"), (11, "amplitude", "CNOT", "probs"), (12, "rz", "CZ", "probs"),
                                            (12, "ry", "CNOT", "expz"), (13, "rz", "CNOT", "expz"), (12, "ry_blocks", "CZ", "probs")])
def test_wide_adjoint_vs_oracle_autograd(n, enc, imp, meas):
    """n = 11..13: psi and lambda in the workspace slabs (qiddm_backward_adjoint_wide) -- weights, angle inputs and
    amplitude-embedded inputs against autograd through the oracle."""
    from oracle import circuits as oc
    from qiddm_amd.circuit import Circuit, execute
    torch.manual_seed(n * 3 + len(enc))
    blocks, layers, B = (2, 2, 3) if enc in ("rz", "ry_blocks") else (1, 3, 3)
    feat = 1500 if enc == "amplitude" else n
    circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=meas, n_rounds=1, n_blocks=blocks,
                   sel_layers=layers, n_features=feat if enc == "amplitude" else 0, pad_with=0.3)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.5)
    x = None if enc == "none" else (torch.rand(B, feat, dtype=torch.float64) + 0.1)
    g = torch.randn(B if x is not None else 1, (1 << n) if meas == "probs" else n, dtype=torch.float64)
    # oracle
    wo = w.clone().requires_grad_(True)
    xo = None if x is None else x.clone().requires_grad_(True)
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure=meas, pad_with=0.3)
    xin = xo if xo is not None else torch.zeros(1, n, dtype=torch.float64)
    out_o = oc.run_circuit(spec, xin, wo)
    (out_o * g).sum().backward()
    # product (float64 kernels)
    wd = w.to("cuda").requires_grad_(True)
    xd = None if x is None else x.to("cuda").requires_grad_(True)
    out_d = execute(circ, xd, wd, "f64", "backprop")
    assert (out_d.cpu() - out_o.detach()).abs().max().item() < 1e-11
    (out_d * g.to("cuda")).sum().backward()
    scale = max(wo.grad.abs().max().item(), 1e-12)
    assert (wd.grad.cpu() - wo.grad).abs().max().item() < 1e-9 * scale + 1e-13
    if xo is not None:
        sx = max(xo.grad.abs().max().item(), 1e-12)
        assert (xd.grad.cpu() - xo.grad).abs().max().item() < 1e-9 * sx + 1e-13


def test_twelve_qubit_qconv_trains_through_the_wide_adjoint():
    """BASELINE config 4's layer shape (C_in = 256, 3x3 -> 2304 features -> 12 wires): weight and input gradients of
    QConv2d against autograd through the oracle (before the wide adjoint existed this raised NotImplementedError)."""
    from oracle import circuits as oc
    from qiddm_amd import circuit as qc
    from qiddm_amd import nn
    torch.manual_seed(4)
    layer = nn.QConv2d(256, 4, qdepth=1).to("cuda").train()
    assert layer.wires == 12
    x = torch.rand(1, 256, 3, 3, dtype=torch.float64)
    g = torch.randn(1, 4, 3, 3, dtype=torch.float64)
    wo = layer.weights.detach().cpu().clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = oc.qconv2d_forward(xo, wo, 4, (3, 3), (1, 1))
    (yo * g).sum().backward()
    prev = qc._default_precision
    qc.set_default_precision("f64")
    try:
        xd = x.to("cuda").requires_grad_(True)
        yd = layer(xd)
        assert (yd.cpu() - yo.detach()).abs().max().item() < 1e-10
        (yd * g.to("cuda")).sum().backward()
    finally:
        qc.set_default_precision(prev)
    sw = max(wo.grad.abs().max().item(), 1e-12)
    assert (layer.weights.grad.cpu() - wo.grad).abs().max().item() < 1e-8 * sw
    sx = max(xo.grad.abs().max().item(), 1e-12)
    assert (xd.grad.cpu() - xo.grad).abs().max().item() < 1e-8 * sx


@pytest.mark.parametrize("c_in,c_out,hw,batch", [(256, 48, (3, 3), 3), (64, 40, (4, 4), 2)])
def test_wide_qconv_trains_through_the_unitary_gemm_route(c_in, c_out, hw, batch, monkeypatch):
    """Layers beyond the thin-product kernel (C4's 12-wire shape: 2304 patch features; 40+ output channels): float32
    training through the circuit unitary with the three backward products as library GEMMs over batch chunks, against
    autograd through the oracle.  The chunk size is forced down so that several chunks (and a ragged last one) run."""
    from oracle import circuits as oc
    from qiddm_amd import circuit as qc
    from qiddm_amd import nn
    torch.manual_seed(9)
    layer = nn.QConv2d(c_in, c_out, qdepth=2).to("cuda").train()
    assert qc.qconv_unitary_route(layer.wires, c_in, (3, 3), c_out) == "gemm"
    monkeypatch.setattr(qc, "_GEMM_CHUNK_BYTES", 2 * hw[0] * hw[1] * c_in * 9 * 4)      # two images per chunk
    x = torch.rand(batch, c_in, *hw, dtype=torch.float64)
    g = torch.randn(batch, c_out, *hw, dtype=torch.float64)
    wo = layer.weights.detach().cpu().clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = oc.qconv2d_forward(xo, wo, c_out, (3, 3), (1, 1))
    (yo * g).sum().backward()
    xd = x.to("cuda").requires_grad_(True)
    yd = layer(xd)
    (yd * g.to("cuda")).sum().backward()
    assert (yd.cpu() - yo.detach()).abs().max().item() < 2e-4
    sw = max(wo.grad.abs().max().item(), 1e-12)
    assert (layer.weights.grad.cpu() - wo.grad).abs().max().item() < 2e-3 * sw
    sx = max(xo.grad.abs().max().item(), 1e-12)
    assert (xd.grad.cpu() - xo.grad).abs().max().item() < 2e-3 * sx


# ---- the pass-structured reverse sweep of the wide CZ family (qsim_wide_cz_adjoint.h) --------------------------------
@pytest.mark.parametrize("n,L,S,meas,B", [(11, 1, 2, "expz", 3), (11, 3, 1, "probs", 2), (12, 2, 2, "probs", 3),
                                          (13, 1, 3, "expz", 2), (13, 5, 1, "expz", 2), (14, 2, 2, "expz", 2),
                                          (16, 1, 2, "expz", 2), (16, 2, 2, "expz", 1)])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 2e-4)])
def test_wide_cz_adjoint_vs_oracle_autograd(n, L, S, meas, B, precision, tol):
    """Weights and angle inputs against autograd through the oracle: even / odd layer counts (the turnaround runs on
    either local-bit set), every layer a block start (S = 1), both read-outs, 11 .. 16 qubits, both precisions."""
    from oracle import circuits as oc
    from qiddm_amd.circuit import Circuit, run_adjoint
    g_ = torch.Generator().manual_seed(1000 * n + 10 * L + S)
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_rounds=1, n_blocks=L, sel_layers=S)
    w = torch.randn(circ.angles_shape, generator=g_, dtype=torch.float64) * 0.6
    x = torch.rand(B, n, generator=g_, dtype=torch.float64) * 2 - 1
    g = torch.randn(B, (1 << n) if meas == "probs" else n, generator=g_, dtype=torch.float64)
    wo, xo = w.clone().requires_grad_(True), x.clone().requires_grad_(True)
    out = oc.run_circuit(oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure=meas), xo, wo)
    (out * g).sum().backward()
    ga, gi = run_adjoint(circ, x.cuda(), w.cuda(), g.cuda(), precision)
    sw = max(wo.grad.abs().max().item(), 1e-12)
    assert (ga.cpu() - wo.grad).abs().max().item() < tol * sw, (ga.cpu() - wo.grad).abs().max().item() / sw
    # one block: the data enters only through the first RZ layer on |0..0>, a global phase (finding F2) -- the input
    # gradient is exactly zero and the oracle's is rounding noise, hence the floor on the scale
    sx = max(xo.grad.abs().max().item(), 1e-3)
    assert (gi.cpu() - xo.grad).abs().max().item() < tol * sx, (gi.cpu() - xo.grad).abs().max().item() / sx
    if L == 1:
        assert gi.abs().max().item() == 0.0


def test_wide_cz_adjoint_many_samples_and_determinism():
    """More samples than resident workgroups (slab pairs and accumulators are reused sample after sample); the summed
    weight gradient equals the sum of per-chunk gradients and is bit-reproducible (fixed-order sums)."""
    from qiddm_amd.circuit import Circuit, run_adjoint
    torch.manual_seed(5)
    n, B = 12, 700
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=1, n_blocks=2, sel_layers=2)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.5).cuda()
    x = (torch.rand(B, n, dtype=torch.float64) * 2 - 1).cuda()
    g = torch.randn(B, n, dtype=torch.float64).cuda()
    ga, gi = run_adjoint(circ, x, w, g, "f64")
    ga2, gi2 = run_adjoint(circ, x, w, g, "f64")
    assert torch.equal(ga, ga2) and torch.equal(gi, gi2)
    parts = [run_adjoint(circ, x[i:i + 175], w, g[i:i + 175], "f64") for i in range(0, B, 175)]
    assert torch.allclose(ga, sum(p[0] for p in parts), atol=1e-9)
    assert torch.allclose(gi, torch.cat([p[1] for p in parts]), atol=1e-12)


# ---- the register-resident reverse sweep of 10-qubit CZ circuits (qsim_cz10_adjoint.h) -------------------------------
@pytest.mark.parametrize("L,S,meas,B", [(1, 2, "probs", 5), (9, 2, "probs", 7), (3, 1, "expz", 4), (2, 9, "expz", 3), (1, 3, "probs", 70)])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 2e-4)])
def test_cz10_adjoint_vs_oracle_autograd(L, S, meas, B, precision, tol):
    """C3's circuit family (differN_noise: 10 wires, RZ re-upload, SEL(CZ), probabilities) and its <Z> sibling: weights and
    angle inputs against autograd through the oracle -- 2 to 18 layers, every layer a block start (S = 1), all nine
    entangler ranges (S = 9), more samples than one wave per workgroup slot (70)."""
    from oracle import circuits as oc
    from qiddm_amd.circuit import Circuit, run_adjoint
    n = 10
    g_ = torch.Generator().manual_seed(10 * L + S)
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_rounds=1, n_blocks=L, sel_layers=S)
    w = torch.randn(circ.angles_shape, generator=g_, dtype=torch.float64) * 0.6
    x = torch.rand(B, n, generator=g_, dtype=torch.float64) * 2 - 1
    g = torch.randn(B, (1 << n) if meas == "probs" else n, generator=g_, dtype=torch.float64)
    wo, xo = w.clone().requires_grad_(True), x.clone().requires_grad_(True)
    out = oc.run_circuit(oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure=meas), xo, wo)
    (out * g).sum().backward()
    ga, gi = run_adjoint(circ, x.cuda(), w.cuda(), g.cuda(), precision)
    ga2, gi2 = run_adjoint(circ, x.cuda(), w.cuda(), g.cuda(), precision)
    assert torch.equal(ga, ga2) and torch.equal(gi, gi2)                     # fixed-order sums
    sw = max(wo.grad.abs().max().item(), 1e-3)
    assert (ga.cpu() - wo.grad).abs().max().item() < tol * sw, (ga.cpu() - wo.grad).abs().max().item() / sw
    sx = max(xo.grad.abs().max().item(), 1e-3)
    assert (gi.cpu() - xo.grad).abs().max().item() < tol * sx, (gi.cpu() - xo.grad).abs().max().item() / sx


# ---- circuits WITHOUT a data encoding (inputs = NULL at the C ABI) on the pass-structured CZ kernels -----------------
@pytest.mark.parametrize("n,L,S,meas,B", [(10, 3, 2, "probs", 5), (10, 1, 3, "expz", 3), (12, 2, 2, "expz", 4),
                                          (12, 3, 1, "probs", 3), (16, 2, 2, "expz", 3), (16, 1, 2, "probs", 2)])
@pytest.mark.parametrize("precision,tol", [("f64", 1e-9), ("f32", 2e-4)])
def test_cz_kernels_without_encoding_forward_and_adjoint(n, L, S, meas, B, precision, tol):
    """`wide_cz_kernel` / `wide_cz_adjoint_kernel` (n = 12, 16) and the 10-qubit `circuit_kernel` / `cz10_adjoint_kernel`
    accept QIDDM_ENC_NONE (inputs = NULL, no block-start re-upload; round-2 advice: every earlier case used "rz").
    Forward and weight gradients against the oracle and autograd through it; every sample is the same circuit, the
    gradient still sums their different upstream rows."""
    from oracle import circuits as oc
    from qiddm_amd.circuit import Circuit, run_adjoint, run_forward
    g_ = torch.Generator().manual_seed(77 * n + 10 * L + S)
    circ = Circuit(n_qubits=n, encoding="none", imprimitive="CZ", measure=meas, n_rounds=1, n_blocks=L, sel_layers=S)
    w = torch.randn(circ.angles_shape, generator=g_, dtype=torch.float64) * 0.6
    cols = (1 << n) if meas == "probs" else n
    g = torch.randn(B, cols, generator=g_, dtype=torch.float64)
    wo = w.clone().requires_grad_(True)
    spec = oc.Spec(n=n, encoding="none", imprimitive="CZ", measure=meas)
    ref = oc.run_circuit(spec, torch.zeros(B, n, dtype=torch.float64), wo)
    (ref * g).sum().backward()
    out = run_forward(circ, None, w.cuda(), precision, batch=B)
    assert out.shape == (B, cols)
    ftol = 1e-11 if precision == "f64" else 2e-5
    assert (out.cpu().double() - ref.detach()).abs().max().item() < ftol
    ga, gi = run_adjoint(circ, None, w.cuda(), g.cuda(), precision)
    assert gi is None
    sw = max(wo.grad.abs().max().item(), 1e-3)
    assert (ga.cpu() - wo.grad).abs().max().item() < tol * sw, (ga.cpu() - wo.grad).abs().max().item() / sw


def test_thin_product_backward_reads_a_channel_slice_of_the_gradient_in_place():
    """dL/dy handed over as one half of a concatenation's gradient (a channel slice: batch stride larger than an image)
    is read in place by the per-pixel-row backward -- bit for bit the gradients of the dense copy."""
    from qiddm_amd import nn
    torch.manual_seed(8)
    layer = nn.QConv2d(16, 8, 1, 0, 2).cuda().train()          # unet_simple's last up-convolution
    x = torch.rand(5, 16, 14, 14, dtype=torch.float64, device="cuda")
    wide = torch.randn(5, 24, 14, 14, dtype=torch.float64, device="cuda")
    for lo in (0, 16):
        gy_view = wide[:, lo:lo + 8]
        assert not gy_view.is_contiguous()
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ga = torch.autograd.grad(layer(xa), [xa, layer.weights], gy_view)
        gb = torch.autograd.grad(layer(xb), [xb, layer.weights], gy_view.contiguous())
        assert torch.equal(ga[0], gb[0]) and torch.equal(ga[1], gb[1])
