"""GPU parity of the n = 11..16 tiled kernel (one workgroup per sample, passes over a workspace
slab) against the CPU oracle -- BASELINE configs 4 (12-qubit QConv2d) and 5 (16-qubit qdense)."""
import pytest
import torch

from oracle import circuits as oc

pytestmark = pytest.mark.gpu

F32_TOL = dict(atol=2e-5, rtol=1e-4)
F64_TOL = dict(atol=1e-11, rtol=1e-10)


def _mk(n, enc, imp, meas, N, L, S, batch, seed, feat=None, pad=0.0, offset=0.0):
    from qiddm_amd.circuit import Circuit
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, L, S, n, 3, generator=g, dtype=torch.float64) * 0.9
    f = feat if feat is not None else n
    x = torch.rand(batch, f, generator=g, dtype=torch.float64) * 2 - 0.5
    if enc == "amplitude":
        x = x.abs()
    circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=meas, n_rounds=N, n_blocks=L,
                   sel_layers=S, n_features=f if enc == "amplitude" else 0, pad_with=pad, enc_offset=offset)
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure=meas, pad_with=pad, enc_offset=offset)
    return circ, spec, x, w


def _run(circ, x, w, precision):
    from qiddm_amd.circuit import run_forward
    out = run_forward(circ, x.cuda(), w.cuda(), precision)
    torch.cuda.synchronize()
    return out.cpu().to(torch.float64)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n", [11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("imp,meas", [("CZ", "expz"), ("CZ", "probs"), ("CNOT", "probs"), ("CNOT", "expz")])
def test_rz_reupload_wide(n, imp, meas, precision):
    batch = 5 if n <= 13 else 2
    circ, spec, x, w = _mk(n, "rz", imp, meas, N=1, L=2, S=3, batch=batch, seed=n * 7)
    got = _run(circ, x, w, precision)
    ref = oc.run_circuit(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n,feat,pad,offset", [(11, 2048, 0.1, 0.0), (12, 2304, 0.5, 0.1), (12, 4096, 0.5, 0.1),
                                              (13, 5000, 0.3, 0.0)])
def test_amplitude_embedding_wide(n, feat, pad, offset, precision):
    """C4: QConv2d with C_in = 256, k = 3 -> 2304 features on 12 wires (nn/qconv.py:24-28)."""
    circ, spec, x, w = _mk(n, "amplitude", "CNOT", "probs", 1, 1, 3, batch=3, seed=n + feat, feat=feat,
                           pad=pad, offset=offset)
    got = _run(circ, x, w, precision)
    ref = oc.run_circuit(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()
    assert torch.allclose(got.sum(1), torch.ones(3, dtype=torch.float64), atol=1e-5)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_ry_and_every_range_wide(precision):
    circ, spec, x, w = _mk(11, "ry", "CNOT", "probs", 1, 1, 12, batch=3, seed=3)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(_run(circ, x, w, precision), oc.run_circuit(spec, x, w), **tol)
    circ, spec, x, w = _mk(12, "rz", "CZ", "expz", 1, 1, 13, batch=3, seed=4)
    assert torch.allclose(_run(circ, x, w, precision), oc.run_circuit(spec, x, w), **tol)


@pytest.mark.parametrize("meas", ["expz", "probs"])
def test_chained_rounds_wide(meas):
    circ, spec, x, w = _mk(11, "rz", "CZ", meas, N=2, L=2, S=2, batch=4, seed=8)
    assert torch.allclose(_run(circ, x, w, "f64"), oc.run_circuit(spec, x, w), **F64_TOL)


def test_c5_shape_16_qubits():
    """C5: (2352, 16, 6, 2)-style LL circuit: 16 wires, 2 rounds x 6 blocks x 2 layers, G = 960."""
    circ, spec, x, w = _mk(16, "rz", "CZ", "expz", N=2, L=6, S=2, batch=2, seed=16)
    assert circ.gate_count() == 960
    got = _run(circ, x, w, "f32")
    ref = oc.run_circuit(spec, x, w)
    assert torch.allclose(got, ref, **F32_TOL), (got - ref).abs().max()


def test_many_samples_grid_stride():
    """More samples than resident workgroups: slabs are reused sample after sample."""
    circ, spec, x, w = _mk(11, "rz", "CZ", "expz", N=1, L=1, S=2, batch=700, seed=1)
    got = _run(circ, x, w, "f32")
    ref = oc.run_circuit(spec, x[:40], w)
    assert torch.allclose(got[:40], ref, **F32_TOL)
    ref_tail = oc.run_circuit(spec, x[-8:], w)
    assert torch.allclose(got[-8:], ref_tail, **F32_TOL)


@pytest.mark.parametrize("enc,imp,meas", [("rz", "CZ", "expz"), ("rz", "CNOT", "probs"), ("amplitude", "CNOT", "probs")])
def test_param_shift_wide(enc, imp, meas):
    from qiddm_amd.circuit import run_shift_sweep
    n = 11
    circ, spec, x, w = _mk(n, enc, imp, meas, 1, 1, 2, batch=3, seed=5, feat=1500 if enc == "amplitude" else None,
                           pad=0.2)
    g = torch.Generator().manual_seed(9)
    gout = torch.randn(3, 2 ** n if meas == "probs" else n, generator=g, dtype=torch.float64)
    wrt_x = enc == "rz"
    ga, gi = run_shift_sweep(circ, x.cuda(), w.cuda(), gout.cuda(), "f64", with_inputs=wrt_x)
    ww = w.clone().requires_grad_(True)
    xx = x.clone().requires_grad_(wrt_x)
    loss = (oc.run_circuit(spec, xx, ww) * gout).sum()
    grads = torch.autograd.grad(loss, [ww, xx] if wrt_x else [ww])
    assert torch.allclose(ga.cpu(), grads[0], atol=1e-9), (ga.cpu() - grads[0]).abs().max()
    if wrt_x:
        assert torch.allclose(gi.cpu(), grads[1][:, :n], atol=1e-9)


# ---- the wide CZ forward (qsim_wide_cz.h): pass structure corner cases --------------------------------------------
@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n", [11, 13, 14, 16])
@pytest.mark.parametrize("L,S", [(1, 1), (1, 2), (3, 1), (1, 3), (2, 2), (5, 1)])
@pytest.mark.parametrize("meas", ["expz", "probs"])
def test_wide_cz_layer_counts(n, L, S, meas, precision):
    """1 layer (the generated product state is measured directly), even / odd layer counts (the last pass runs on
    either local-bit set), every layer a block start (S = 1: data re-upload in every diagonal)."""
    circ, spec, x, w = _mk(n, "rz", "CZ", meas, N=1, L=L, S=S, batch=3, seed=100 * n + 10 * L + S)
    got = _run(circ, x, w, precision)
    ref = oc.run_circuit(spec, x, w)
    tol = F64_TOL if precision == "f64" else F32_TOL
    assert torch.allclose(got, ref, **tol), (got - ref).abs().max()


@pytest.mark.parametrize("n,meas", [(12, "expz"), (14, "probs"), (16, "expz")])
def test_wide_cz_chained_rounds_and_scaled_encoding(n, meas):
    """Three chained rounds (x <- out[:, :n]) and RZ(pi/2 * x) encoding (QIDDM_A_differN_basePL, nn/qdense.py:2215)."""
    import math
    from qiddm_amd.circuit import Circuit
    g = torch.Generator().manual_seed(n)
    w = torch.randn(3, 2, 2, n, 3, generator=g, dtype=torch.float64) * 0.8
    x = torch.rand(2, n, generator=g, dtype=torch.float64) * 2 - 1
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_rounds=3, n_blocks=2, sel_layers=2,
                   enc_scale=math.pi / 2)
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure=meas, enc_scale=math.pi / 2)
    for precision, tol in (("f64", F64_TOL), ("f32", dict(atol=5e-5, rtol=2e-4))):
        got = _run(circ, x, w, precision)
        ref = oc.run_circuit(spec, x, w)
        assert torch.allclose(got, ref, **tol), (precision, (got - ref).abs().max())


@pytest.mark.parametrize("n,batch", [(12, 1300), (16, 37)])
def test_wide_cz_slab_reuse_and_ragged_batches(n, batch):
    """More samples than resident workgroups (slabs are reused sample after sample), a batch that is not a
    multiple of anything; probability rows sum to one at full size, spot rows against the oracle."""
    circ, spec, x, w = _mk(n, "rz", "CZ", "probs", N=1, L=2, S=2, batch=batch, seed=n)
    got = _run(circ, x, w, "f32")
    assert got.shape == (batch, 2 ** n)
    assert torch.allclose(got.sum(1), torch.ones(batch, dtype=torch.float64), atol=2e-5)
    idx = [0, 1, batch // 2, batch - 1]
    ref = oc.run_circuit(spec, x[idx], w)
    assert torch.allclose(got[idx], ref, **F32_TOL), (got[idx] - ref).abs().max()
    assert _run(circ, x[:0], w, "f32").shape == (0, 2 ** n)
