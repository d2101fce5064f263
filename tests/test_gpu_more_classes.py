"""GPU checks of the RY re-uploading encoding (QIDDM_ENC_RY_BLOCKS) on every kernel and of the
remaining reference classes (qiddm_amd/nn/qdense_more.py) against oracle compositions."""
import math

import pytest
import torch

from oracle import circuits as oc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(n, L, S, batch, seed, meas="expz", imp="CZ"):
    from qiddm_amd.circuit import Circuit
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(1, L, S, n, 3, generator=g, dtype=torch.float64) * 0.8
    x = torch.rand(batch, n, generator=g, dtype=torch.float64) * 2 - 0.5
    circ = Circuit(n_qubits=n, encoding="ry_blocks", imprimitive=imp, measure=meas, n_blocks=L, sel_layers=S)
    spec = oc.Spec(n=n, encoding="ry_blocks", imprimitive=imp, measure=meas)
    gout = torch.randn(batch, n if meas == "expz" else 2 ** n, generator=g, dtype=torch.float64)
    return circ, spec, x, w, gout


@pytest.mark.parametrize("n,meas,imp", [(3, "expz", "CZ"), (6, "probs", "CNOT"), (8, "expz", "CZ"), (10, "probs", "CZ"),
                                       (11, "expz", "CZ"), (12, "probs", "CNOT")])
def test_ry_reupload_forward_and_gradients(n, meas, imp):
    from qiddm_amd.circuit import run_adjoint, run_forward, run_shift_sweep
    circ, spec, x, w, gout = _mk(n, 3, 2, 5, n, meas, imp)
    got = run_forward(circ, x.cuda(), w.cuda(), "f64").cpu()
    assert torch.allclose(got, oc.run_circuit(spec, x, w), atol=1e-11)
    assert circ.gate_count() == oc.gate_count(spec, 1, 3, 2)
    ww = w.clone().requires_grad_(True)
    xx = x.clone().requires_grad_(True)
    ra, ri = torch.autograd.grad((oc.run_circuit(spec, xx, ww) * gout).sum(), [ww, xx])
    ga, gi = run_shift_sweep(circ, x.cuda(), w.cuda(), gout.cuda(), "f64")
    assert torch.allclose(ga.cpu(), ra, atol=1e-9) and torch.allclose(gi.cpu(), ri, atol=1e-9)
    if n <= 10:
        ga, gi = run_adjoint(circ, x.cuda(), w.cuda(), gout.cuda(), "f64")
        assert torch.allclose(ga.cpu(), ra, atol=1e-9) and torch.allclose(gi.cpu(), ri, atol=1e-9)


def _img(b, w, seed):
    return torch.rand(b, 1, w, w, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)


def test_qiddm_pl_noise1_rounds():
    from qiddm_amd import nn
    torch.manual_seed(1)
    m = nn.QIDDM_PL_noise1(64, 4, 3, 2).to(DEV, dtype=torch.double)
    red = torch.randn(7, 4, dtype=torch.float64)
    got = m.quantum_rounds(red.to(DEV)).cpu()
    ref = oc.run_circuit(oc.Spec(n=4, encoding="ry_blocks", imprimitive="CZ", measure="expz"), red,
                         m.weights1.detach().cpu())
    assert torch.allclose(got, ref, atol=5e-5)
    y = m(_img(7, 8, 2).to(DEV))
    assert y.shape == (7, 1, 8, 8)


def test_bias_false_and_ll_old_and_l_b():
    from qiddm_amd import nn
    torch.manual_seed(2)
    x = _img(6, 8, 3)
    m = nn.QIDDM_bias_false(64, 4, 2, 2).to(DEV, dtype=torch.double)
    assert tuple(m.weights1.shape) == (2, 2, 3, 4, 3)
    got = m(x.to(DEV)).detach().cpu()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    red = x.reshape(6, -1) @ sd["linear_down.weight"].T
    ev = oc.run_circuit(oc.Spec(n=4, encoding="rz", imprimitive="CZ", measure="expz"), red, sd["weights1"])
    assert torch.allclose(got, (ev @ sd["linear_up.weight"].T).reshape(6, 1, 8, 8), atol=5e-5)
    ll = nn.QIDDM_LL_old(64, 4, 2, 2).to(DEV, dtype=torch.double)
    with torch.no_grad():
        a = ll(x.to(DEV)).cpu()                      # fused launch
    b = ll(x.to(DEV)).detach().cpu()                 # per-round QNode path
    assert torch.allclose(a, b, atol=1e-4)
    lb = nn.QIDDM_L_B(64, 4, 2, 2).to(DEV, dtype=torch.double).train()
    out = lb(x.to(DEV))
    out.square().mean().backward()
    assert lb.weights1.grad is not None and lb.linear_down.weight.grad is not None   # not detached here
    sd = {k: v.detach().cpu() for k, v in lb.state_dict().items()}
    red = x.reshape(6, -1) @ sd["linear_down.weight"].T + sd["linear_down.bias"]
    for n in range(2):
        red = (red - red.mean(0)) / torch.sqrt(red.var(0, unbiased=False) + 1e-5) * sd["batchnorm.weight"] + sd["batchnorm.bias"]
        red = oc.run_round(oc.Spec(n=4, encoding="rz", imprimitive="CZ", measure="expz"), red, sd["weights1"][n])
    ref = (red @ sd["linear_up.weight"].T + sd["linear_up.bias"]).reshape(6, 1, 8, 8)
    assert torch.allclose(out.detach().cpu(), ref, atol=5e-5)


def test_probs_chained_variants():
    from qiddm_amd import nn
    torch.manual_seed(3)
    spec = oc.Spec(n=6, encoding="rz", imprimitive="CZ", measure="probs")
    red = torch.randn(5, 6) * 1.2
    m = nn.differN_new_pca(8, 2, 2).to(DEV)
    got = m.forward_from_reduced(red.to(DEV)).detach().cpu()
    x = red.double()
    for n in range(2):                                # post-processing BETWEEN the rounds (:806-809)
        x = oc.post_process_dense(oc.run_round(spec, x, m.weights.detach().cpu()[n]), 64)
    assert torch.allclose(got.reshape(5, 64), x, atol=2e-3)
    s = nn.QIDDM_A_sameN(8, 2, 3).to(DEV)
    img = _img(4, 8, 5)
    got = s(img.to(DEV)).detach().cpu()
    p = img.reshape(4, 64)
    for _ in range(3):                                # one shared weight tensor for all rounds
        p = oc.run_round(spec, p, s.weights.detach().cpu())
    assert torch.allclose(got.reshape(4, 64), oc.post_process_dense(p, 64), atol=2e-3)
    a = nn.QIDDM_A_differN_basePL(8, 2, 2).to(DEV)
    got = a.forward_from_reduced(red.double().to(DEV)).cpu()
    spec_s = oc.Spec(n=6, encoding="rz", imprimitive="CZ", measure="probs", enc_scale=math.pi / 2)
    x = red.double()
    for n in range(2):
        x = oc.post_process_dense(oc.run_round(spec_s, x, a.weights1.detach().cpu()[n]), 64)
    assert torch.allclose(got, x, atol=2e-3)
    assert nn.QIDDM_A_differN_NEW(8, 2, 2).save_name() == "QIDDM_pca_new=6_L=2_N=2"


def test_front_end_variants_run():
    from qiddm_amd import nn
    torch.manual_seed(4)
    x = _img(12, 8, 6)
    for ctor, args in [(nn.differN_old_conv, (8, 2, 2)), (nn.differN_new_conv, (8, 2, 2)), (nn.QIDDM_CL_new, (64, 4, 2, 1)),
                       (nn.QIDDM_CL_old, (64, 4, 2, 1)), (nn.QIDDM_PL_old, (64, 4, 2, 1)), (nn.QIDDM_PP_noise, (64, 4, 2, 1)),
                       (nn.QIDDM_PP_old, (64, 4, 2, 1)), (nn.QIDDM_A_differN_NEW, (8, 2, 1))]:
        m = ctor(*args).to(DEV, dtype=torch.double)
        y = m(x.to(DEV))
        assert y.shape == x.shape and torch.isfinite(y).all(), ctor.__name__
    cl = nn.QIDDM_CL_new(64, 4, 2, 1).to(DEV, dtype=torch.double)
    sd = {k: v.detach().cpu() for k, v in cl.state_dict().items()}
    red = torch.nn.functional.conv2d(x, sd["conv_layer.weight"], sd["conv_layer.bias"], stride=2, padding=1)
    red = red.reshape(12, 4, -1).mean(2)
    ev = oc.run_circuit(oc.Spec(n=4, encoding="rz", imprimitive="CZ", measure="expz"), red, sd["weights1"])
    ref = (ev @ sd["linear_up.weight"].T + sd["linear_up.bias"]).reshape(12, 1, 8, 8)
    with torch.no_grad():
        assert torch.allclose(cl(x.to(DEV)).cpu(), ref, atol=5e-5)
