"""Known-answer tests KA1-KA12 that pin the CPU oracle
without PennyLane, plus the (a)-vs-(b) cross-check between the two independent
oracle implementations."""
import math

import numpy as np
import pytest
import torch

from oracle import circuits as oc
from oracle import dense as od
from oracle import statevector as sv

torch.manual_seed(0)


def _rand_w(*shape, scale=0.4, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


# KA1 ------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 4, 6])
def test_ka1_zero_angles_identity(n):
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="probs")
    x = torch.rand(3, n, dtype=torch.float64)
    w = torch.zeros(1, 2, 3, n, 3, dtype=torch.float64)
    p = oc.run_circuit(spec, x, w)
    e0 = torch.zeros(3, 2 ** n, dtype=torch.float64)
    e0[:, 0] = 1
    assert torch.allclose(p, e0, atol=1e-14)
    spec.measure = "expz"
    ev = oc.run_circuit(spec, x, w)
    assert torch.allclose(ev, torch.ones(3, n, dtype=torch.float64), atol=1e-14)


# KA2 ------------------------------------------------------------------------
def test_ka2_single_qubit_rot():
    phi, theta, omega = 0.3, 1.1, -0.7
    u = sv.rot_matrix(phi, theta, omega)
    st = sv.apply_1q(sv.zero_state(1, 1), u, 0, 1)
    assert abs(sv.probs(st)[0, 1].item() - math.sin(theta / 2) ** 2) < 1e-15
    assert abs(sv.expval_z(st, 1)[0, 0].item() - math.cos(theta)) < 1e-15
    # explicit matrix form quoted in SURVEY section 8c item (2)
    c, s = math.cos(theta / 2), math.sin(theta / 2)
    ref = np.array([[np.exp(-0.5j * (phi + omega)) * c, -np.exp(0.5j * (phi - omega)) * s],
                    [np.exp(-0.5j * (phi - omega)) * s, np.exp(0.5j * (phi + omega)) * c]])
    assert np.allclose(u.numpy(), ref, atol=1e-15)
    assert np.allclose(od.rot(phi, theta, omega), ref, atol=1e-15)


# KA3 ------------------------------------------------------------------------
def test_ka3_wire_order():
    st = sv.apply_1q(sv.zero_state(1, 2), sv.ry_matrix(math.pi), 0, 2)
    p = sv.probs(st)[0]
    assert abs(p[2].item() - 1) < 1e-15          # |10>: wire 0 is the most significant bit
    ev = sv.expval_z(st, 2)[0]
    assert abs(ev[0].item() + 1) < 1e-15 and abs(ev[1].item() - 1) < 1e-15


# KA4 ------------------------------------------------------------------------
def test_ka4_norm_preserved_deep():
    n = 6
    w = _rand_w(1, 1, 60, n, 3)
    x = torch.rand(2, 40, dtype=torch.float64)
    spec = oc.Spec(n=n, encoding="amplitude", imprimitive="CNOT", measure="probs", pad_with=0.1)
    p = oc.run_circuit(spec, x, w)
    assert torch.allclose(p.sum(1), torch.ones(2, dtype=torch.float64), atol=1e-12)


# KA5 (finding F2) -------------------------------------------------------------
def test_ka5_qnn_output_independent_of_input():
    n = 4
    w = _rand_w(1, 1, 3, n, 3)
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz")
    a = oc.run_circuit(spec, torch.rand(5, n, dtype=torch.float64) * 3, w)
    b = oc.run_circuit(spec, torch.zeros(5, n, dtype=torch.float64), w)
    assert torch.allclose(a, b, atol=1e-13)
    # re-uploading circuits DO depend on the input from block 2 on
    w2 = _rand_w(1, 3, 2, n, 3)
    a = oc.run_circuit(spec, torch.rand(5, n, dtype=torch.float64) * 3, w2)
    b = oc.run_circuit(spec, torch.zeros(5, n, dtype=torch.float64), w2)
    assert not torch.allclose(a, b, atol=1e-3)


# KA6 ------------------------------------------------------------------------
def test_ka6_two_qubit_rings():
    n = 2
    st = torch.randn(1, 4, dtype=torch.complex128)
    w0 = torch.zeros(1, n, 3, dtype=torch.float64)
    assert torch.allclose(sv.strongly_entangling_layers(st, w0, n, "CZ"), st)
    out = sv.strongly_entangling_layers(st, w0, n, "CNOT")
    ref = sv.apply_cnot(sv.apply_cnot(st, 0, 1, n), 1, 0, n)
    assert torch.allclose(out, ref)
    # CNOT(0,1): |10> -> |11>
    e = torch.zeros(1, 4, dtype=torch.complex128)
    e[0, 2] = 1
    assert sv.apply_cnot(e, 0, 1, n)[0, 3] == 1


# KA7 ------------------------------------------------------------------------
@pytest.mark.parametrize("n", [3, 5])
def test_ka7_cnot_ring_permutation(n):
    w0 = torch.zeros(1, 1, 1, n, 3, dtype=torch.float64)
    spec = oc.Spec(n=n, encoding="amplitude", imprimitive="CNOT", measure="probs", pad_with=0.0)
    for j in range(2 ** n):
        feat = torch.zeros(1, 2 ** n, dtype=torch.float64)
        feat[0, j] = 1
        p = oc.run_circuit(spec, feat, w0)
        # walk the ring by hand on the basis label
        bits = [(j >> (n - 1 - w)) & 1 for w in range(n)]
        for i in range(n):
            t = (i + 1) % n
            bits[t] ^= bits[i]
        k = sum(b << (n - 1 - w) for w, b in enumerate(bits))
        assert p[0, k].item() == pytest.approx(1.0)


# KA8 ------------------------------------------------------------------------
def test_ka8_amplitude_embedding():
    a = sv.amplitude_embedding(torch.tensor([[3.0, 4.0]]), 2, pad_with=0.0)
    assert torch.allclose(a.real, torch.tensor([[0.6, 0.8, 0, 0]], dtype=torch.float64))
    a = sv.amplitude_embedding(torch.tensor([[3.0, 4.0]]), 2, pad_with=0.1)
    ref = torch.tensor([[3, 4, 0.1, 0.1]], dtype=torch.float64) / math.sqrt(25.02)
    assert torch.allclose(a.real, ref, atol=1e-15)
    with pytest.raises(ValueError):
        sv.amplitude_embedding(torch.ones(1, 5), 2, pad_with=0.1)
    with pytest.raises(ValueError):
        sv.amplitude_embedding(torch.ones(1, 3), 2, pad_with=None)
    assert np.allclose(od.amp_embed(np.array([3.0, 4.0]), 2, 0.1), ref.numpy()[0])


# KA9: implementation (a) vs dense Kronecker implementation (b) ----------------
@pytest.mark.parametrize("n,imp", [(1, "CZ"), (2, "CNOT"), (3, "CZ"), (4, "CNOT"), (5, "CZ"),
                                   (6, "CNOT"), (7, "CZ"), (8, "CNOT")])
def test_ka9_strided_vs_dense(n, imp):
    s_layers = 2 if n < 7 else n  # n layers exercise every range incl. wrap-around
    w = _rand_w(s_layers, n, 3, scale=1.0, seed=n)
    u = od.sel_unitary(w.numpy(), n, imp)
    assert np.allclose(u.conj().T @ u, np.eye(2 ** n), atol=1e-12)
    rng = np.random.default_rng(n)
    psi = rng.normal(size=2 ** n) + 1j * rng.normal(size=2 ** n)
    psi /= np.linalg.norm(psi)
    ref = u @ psi
    out = sv.strongly_entangling_layers(torch.from_numpy(psi).unsqueeze(0), w, n, imp)
    assert np.allclose(out[0].numpy(), ref, atol=1e-12)


@pytest.mark.parametrize("enc", ["rz", "ry", "amplitude"])
@pytest.mark.parametrize("meas", ["probs", "expz"])
def test_ka9_full_templates_vs_dense(enc, meas):
    n, L, S = 4, 2, 2
    imp = "CNOT" if enc == "amplitude" else "CZ"
    w = _rand_w(1, L, S, n, 3, scale=0.9, seed=5)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, size=(3, 9 if enc == "amplitude" else n))
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure=meas, pad_with=0.3, enc_scale=1.0)
    got = oc.run_circuit(spec, torch.from_numpy(x), w).numpy()
    for b in range(3):
        if enc == "amplitude":
            psi = od.amp_embed(x[b], n, 0.3)
        else:
            psi = np.zeros(2 ** n, dtype=np.complex128)
            psi[0] = 1
        for blk in range(L):
            if enc == "rz":
                psi = od.rz_layer_unitary(x[b], n) @ psi
            elif enc == "ry" and blk == 0:
                psi = od.ry_layer_unitary(x[b], n) @ psi
            psi = od.sel_unitary(w[0, blk].numpy(), n, imp) @ psi
        ref = od.probs(psi) if meas == "probs" else od.expval_z(psi, n)
        assert np.allclose(got[b], ref, atol=1e-12)


# KA10: parameter shift == autograd == finite differences ----------------------
def test_ka10_parameter_shift_matches_autograd():
    n, L, S = 3, 2, 2
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz")
    w = _rand_w(1, L, S, n, 3, seed=3).requires_grad_(True)
    x = torch.rand(2, n, dtype=torch.float64)
    g = torch.randn(2, n, dtype=torch.float64)
    loss = (oc.run_circuit(spec, x, w) * g).sum()
    (auto,) = torch.autograd.grad(loss, w)
    flat = w.detach().reshape(-1)
    ps = torch.zeros_like(flat)
    fd = torch.zeros_like(flat)
    for i in range(flat.numel()):
        for sgn in (+1, -1):
            sh = flat.clone()
            sh[i] += sgn * math.pi / 2
            ps[i] += sgn * 0.5 * (oc.run_circuit(spec, x, sh.reshape(w.shape)) * g).sum()
            sh = flat.clone()
            sh[i] += sgn * 1e-6
            fd[i] += sgn * (oc.run_circuit(spec, x, sh.reshape(w.shape)) * g).sum() / 2e-6
    assert torch.allclose(ps, auto.reshape(-1), atol=1e-10)
    assert torch.allclose(fd, auto.reshape(-1), atol=1e-7)


# KA11 ------------------------------------------------------------------------
def test_ka11_differn_chaining():
    n, L = 4, 2
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="probs")
    w = _rand_w(2, L, 2, n, 3, seed=9)
    x = torch.rand(3, n, dtype=torch.float64)
    r1 = oc.run_round(spec, x, w[0])
    r2 = oc.run_round(spec, r1[:, :n], w[1])   # only columns 0..n-1 are read (nn/qdense.py:427)
    assert torch.allclose(oc.run_circuit(spec, x, w), r2, atol=1e-15)


# KA12 ------------------------------------------------------------------------
def test_ka12_post_process():
    p = torch.tensor([[0.001, 0.5, 0.0005, 0.4985]], dtype=torch.float64)
    out = oc.post_process_dense(p, 3)
    assert torch.allclose(out, torch.tensor([[0.003, 1.0, 0.0015]], dtype=torch.float64))
    # qconv: clamp(p*D/2,0,1)[:, ::2][:, :C_out]
    x = torch.rand(2, 1, 5, 5, dtype=torch.float64)
    w = _rand_w(2, 4, 3)
    y = oc.qconv2d_forward(x, w, out_channels=8)
    assert y.shape == (2, 8, 5, 5) and y.min() >= 0 and y.max() <= 1


def test_gate_counts_match_survey():
    # QNN_noise(784,8,14) G=232; LL(784,8,6,2) G=480; differN(28,9,2) G=900;
    # QDenseUndirected_old_noise(60,28) G=1201; QNN_noise(64,4,2) G=20
    assert oc.gate_count(oc.Spec(8, "rz"), 1, 1, 14) == 232
    assert oc.gate_count(oc.Spec(8, "rz"), 2, 6, 2) == 480
    assert oc.gate_count(oc.Spec(10, "rz"), 2, 9, 2) == 900
    assert oc.gate_count(oc.Spec(10, "amplitude"), 1, 1, 60) == 1201
    assert oc.gate_count(oc.Spec(4, "rz"), 1, 1, 2) == 20
