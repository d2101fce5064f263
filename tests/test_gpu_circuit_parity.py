"""GPU parity: the HIP statevector engine (through the C ABI) against the CPU
oracle on identical seeded inputs.

Tolerances: f64 kernels 1e-11 absolute (the reference's
own precision); f32 kernels atol 2e-5 / rtol 1e-4 on raw probabilities and
expectation values for up to ~1200 sequential gates.
"""
import math

import pytest
import torch

from oracle import circuits as oc

pytestmark = pytest.mark.gpu

F32_TOL = dict(atol=2e-5, rtol=1e-4)
F64_TOL = dict(atol=1e-11, rtol=1e-10)


def _mk(n, enc, imp, meas, N, L, S, batch, seed, feat=None, scale=1.0, offset=0.0, pad=0.0):
    from qiddm_amd.circuit import Circuit
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, L, S, n, 3, generator=g, dtype=torch.float64) * 0.9
    f = feat if feat is not None else n
    x = torch.rand(batch, f, generator=g, dtype=torch.float64) * 2 - 0.5
    circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=meas, n_rounds=N, n_blocks=L,
                   sel_layers=S, n_features=f if enc == "amplitude" else 0, enc_scale=scale,
                   enc_offset=offset, pad_with=pad)
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure=meas, enc_scale=scale,
                   enc_offset=offset, pad_with=pad)
    return circ, spec, x, w


def _run(circ, x, w, precision):
    from qiddm_amd.circuit import run_forward
    out = run_forward(circ, x.cuda(), w.cuda(), precision)
    torch.cuda.synchronize()
    return out.cpu().to(torch.float64)


def _oracle(spec, x, w, scale_chain=1.0):
    return oc.run_circuit(spec, x, w)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n", list(range(1, 11)))
@pytest.mark.parametrize("imp,meas", [("CZ", "expz"), ("CZ", "probs"), ("CNOT", "probs"), ("CNOT", "expz")])
def test_rz_reupload_family(n, imp, meas, precision):
    """Rows A1-A3: RZ data re-uploading + SEL + probs/<Z>; ragged batch."""
    circ, spec, x, w = _mk(n, "rz", imp, meas, N=1, L=2, S=min(3, max(n, 1)), batch=37, seed=100 + n)
    got = _run(circ, x, w, precision)
    ref = _oracle(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n", [2, 4, 6, 7, 8, 10])
def test_every_entangler_range(n, precision):
    """S = 2n-1 layers walk through every range r = 1..n-1 and wrap around."""
    for imp in ("CZ", "CNOT"):
        circ, spec, x, w = _mk(n, "rz", imp, "probs", N=1, L=1, S=2 * n - 1, batch=5, seed=7 * n)
        got = _run(circ, x, w, precision)
        ref = _oracle(spec, x, w)
        tol = F64_TOL if precision == "f64" else F32_TOL
        assert torch.allclose(got, ref, **tol), (imp, (got - ref).abs().max())


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n,feat,pad,offset", [(1, 2, 0.1, 0.0), (2, 3, 0.1, 0.0), (4, 9, 0.5, 0.1),
                                              (6, 64, 0.1, 0.0), (6, 50, 0.1, 0.0), (7, 72, 0.5, 0.1),
                                              (8, 200, 0.3, 0.0), (10, 784, 0.1, 0.0),
                                              (10, 1024, 0.1, 0.0)])
def test_amplitude_embedding_family(n, feat, pad, offset, precision):
    """Rows A4/A5: AmplitudeEmbedding(pad_with, normalize) + SEL(CNOT) + probs."""
    circ, spec, x, w = _mk(n, "amplitude", "CNOT", "probs", N=1, L=1, S=3, batch=19, seed=n + feat,
                           feat=feat, pad=pad, offset=offset)
    x = x.abs()
    got = _run(circ, x, w, precision)
    ref = _oracle(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()
    assert torch.allclose(got.sum(1), torch.ones(19, dtype=torch.float64), atol=1e-5)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n", [3, 6, 9])
def test_ry_angle_embedding(n, precision):
    """QNN_A: AngleEmbedding(rotation='Y') + SEL(CNOT) + probs (nn/qdense.py:164-183)."""
    circ, spec, x, w = _mk(n, "ry", "CNOT", "probs", N=1, L=1, S=4, batch=11, seed=n)
    got = _run(circ, x, w, precision)
    ref = _oracle(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n,meas", [(4, "expz"), (8, "expz"), (6, "probs"), (10, "probs"), (3, "probs")])
def test_chained_rounds_fused(n, meas, precision):
    """N chained QNode rounds in one launch (nn/qdense.py:464-465, 1631-1635; KA11)."""
    circ, spec, x, w = _mk(n, "rz", "CZ", meas, N=3, L=2, S=2, batch=9, seed=n)
    got = _run(circ, x, w, precision)
    ref = _oracle(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()


def test_scaled_rz_encoding():
    """RZ(pi/2 * x) encoding of QIDDM_A_differN_basePL (nn/qdense.py:2215)."""
    circ, spec, x, w = _mk(5, "rz", "CZ", "probs", N=1, L=3, S=2, batch=7, seed=2, scale=math.pi / 2)
    got = _run(circ, x, w, "f64")
    assert torch.allclose(got, _oracle(spec, x, w), **F64_TOL)


def test_known_answers_on_device():
    from qiddm_amd.circuit import Circuit
    # KA1: zero angles -> |0..0>
    circ = Circuit(n_qubits=8, encoding="rz", imprimitive="CZ", measure="probs", n_blocks=2, sel_layers=2)
    w = torch.zeros(circ.angles_shape, dtype=torch.float64)
    p = _run(circ, torch.rand(4, 8, dtype=torch.float64), w, "f32")
    e0 = torch.zeros(4, 256, dtype=torch.float64)
    e0[:, 0] = 1
    assert torch.allclose(p, e0, atol=1e-6)
    # KA3: wire order -- theta = pi on wire 0 of n=2 -> |10> = index 2
    circ = Circuit(n_qubits=2, encoding="rz", imprimitive="CZ", measure="probs")
    w = torch.zeros(circ.angles_shape, dtype=torch.float64)
    w[0, 0, 0, 0, 1] = math.pi
    p = _run(circ, torch.zeros(1, 2, dtype=torch.float64), w, "f64")
    assert abs(p[0, 2].item() - 1) < 1e-12
    # KA5 / F2: QNN output independent of its input
    circ = Circuit(n_qubits=8, encoding="rz", imprimitive="CZ", measure="expz", sel_layers=14)
    w = torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4
    a = _run(circ, torch.rand(6, 8, dtype=torch.float64) * 5, w, "f64")
    b = _run(circ, torch.zeros(6, 8, dtype=torch.float64), w, "f64")
    assert torch.allclose(a, b, atol=1e-12)


@pytest.mark.parametrize("batch", [0, 1, 2, 63, 64, 65, 1000])
def test_ragged_batches(batch):
    circ, spec, x, w = _mk(4, "rz", "CZ", "expz", N=1, L=2, S=2, batch=max(batch, 1), seed=batch)
    x = x[:batch]
    got = _run(circ, x, w, "f32")
    assert got.shape == (batch, 4)
    if batch:
        assert torch.allclose(got, _oracle(spec, x, w), **F32_TOL)


def test_deep_circuit_f32_drift():
    """QDenseUndirected_old_noise(60, 28): 1201 sequential gates at n=10 (row A4)."""
    circ, spec, x, w = _mk(10, "amplitude", "CNOT", "probs", N=1, L=1, S=60, batch=4, seed=5,
                           feat=784, pad=0.1)
    x = x.abs()
    got = _run(circ, x, w, "f32")
    ref = _oracle(spec, x, w)
    assert torch.allclose(got, ref, **F32_TOL), (got - ref).abs().max()
    assert torch.allclose(got.sum(1), torch.ones(4, dtype=torch.float64), atol=1e-5)   # KA4


def test_noncontiguous_and_wide_inputs():
    """Only the first n columns are read (nn/qdense.py:427); row stride honoured."""
    circ, spec, x, w = _mk(6, "rz", "CZ", "probs", N=1, L=2, S=2, batch=10, seed=1, feat=64)
    got = _run(circ, x, w, "f64")
    ref = _oracle(spec, x[:, :6], w)
    assert torch.allclose(got, ref, **F64_TOL)


def test_errors_are_loud():
    from qiddm_amd._capi import QiddmError
    from qiddm_amd.circuit import Circuit, run_forward
    w = torch.zeros(1, 1, 1, 17, 3, dtype=torch.float64).cuda()
    with pytest.raises(QiddmError, match="exceeds the limit 16"):
        run_forward(Circuit(n_qubits=17, encoding="rz"), torch.zeros(2, 17).cuda(), w)
    with pytest.raises(ValueError, match="Features must be of length"):
        run_forward(Circuit(n_qubits=2, encoding="amplitude", imprimitive="CNOT", measure="probs"),
                    torch.zeros(2, 5).cuda(), torch.zeros(1, 1, 1, 2, 3, dtype=torch.float64).cuda())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        run_forward(Circuit(n_qubits=2), torch.zeros(2, 2), torch.zeros(1, 1, 1, 2, 3))


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n,meas,N,L,S", [(8, "expz", 1, 1, 14), (8, "probs", 2, 3, 2), (9, "expz", 2, 2, 2), (9, "probs", 1, 3, 1),
                                          (10, "probs", 2, 3, 2), (10, "expz", 1, 2, 9)])
def test_lean_folded_forward_large_batches(n, meas, N, L, S, precision):
    """More than one wave of work per SIMD (> 1024 sample groups) sends the forward of a CZ circuit with RZ encoding
    through circuit_folded_kernel (scalar-register tables, 2-4 waves per SIMD): chained rounds, every entangler
    range (S = 9 at n = 10, 14 layers at n = 8), both read-outs, a ragged batch; 48 rows spread over the batch against
    the oracle, all rows normalised."""
    batch = 1500 + n
    circ, spec, x, w = _mk(n, "rz", "CZ", meas, N=N, L=L, S=S, batch=batch, seed=7000 + 10 * n + S)
    got = _run(circ, x, w, precision)
    idx = torch.linspace(0, batch - 1, 48).long()
    ref = _oracle(spec, x[idx], w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(got[idx], ref, **tol), (got[idx] - ref).abs().max()
    if meas == "probs":
        assert torch.allclose(got.sum(1), torch.ones(batch, dtype=torch.float64), atol=1e-5)
    else:
        assert got.abs().max() <= 1 + 1e-5


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("n,enc,imp,N,L,S,B,cols", [(10, "rz", "CZ", 2, 9, 2, 1024, 784),      # C3's differN_noise(28, 9, 2)
                                                    (10, "rz", "CZ", 1, 2, 2, 37, 1024),
                                                    (10, "rz", "CZ", 2, 3, 2, 3000, 784),      # > 1024 sample groups
                                                    (6, "rz", "CZ", 2, 4, 2, 50, 64),
                                                    (8, "rz", "CZ", 1, 3, 2, 5000, 200),       # the lean folded kernel at n = 8
                                                    (10, "amplitude", "CNOT", 1, 1, 5, 33, 784),  # QDenseUndirected family
                                                    (3, "amplitude", "CNOT", 1, 1, 2, 9, 5)])
def test_forward_post_is_the_post_processed_forward(n, enc, imp, N, L, S, B, cols, precision):
    """qiddm_forward_post == clamp(float64(qiddm_forward)[:, :cols] * cols, 0, 1), bit for bit (the same kernels with the
    reference's `_post_process`, nn/qdense.py:49-54 / 443-448, fused into the store), and == the oracle."""
    from qiddm_amd.circuit import Circuit, run_forward, run_forward_post
    g = torch.Generator().manual_seed(n * 100 + B)
    feats = cols if enc == "amplitude" else n
    circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure="probs", n_rounds=N, n_blocks=L, sel_layers=S,
                   n_features=feats if enc == "amplitude" else 0, pad_with=0.1)
    w = torch.randn(circ.angles_shape, generator=g, dtype=torch.float64) * 0.5
    x = torch.rand(B, feats, generator=g, dtype=torch.float64) if enc == "amplitude" else \
        torch.randn(B, n, generator=g, dtype=torch.float64)
    raw = run_forward(circ, x.cuda(), w.cuda(), precision)
    want = torch.clamp(raw.double()[:, :cols] * float(cols), 0, 1)
    got = run_forward_post(circ, x.cuda(), w.cuda(), cols, float(cols), precision)
    assert got.dtype == torch.float64 and got.shape == (B, cols)
    assert torch.equal(got, want)
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure="probs", pad_with=0.1)
    ref = torch.clamp(oc.run_circuit(spec, x[:64], w)[:, :cols] * cols, 0, 1)
    tol = 1e-9 if precision == "f64" else 2e-5 * cols
    assert (got[:64].cpu() - ref).abs().max().item() < tol
