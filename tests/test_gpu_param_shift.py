"""GPU parity of the parameter-shift backward (KA10): the HIP sweep re-invokes
the forward kernel for every +-pi/2 replica; its gradients must equal torch
autograd through the CPU oracle (what PennyLane's ``backprop`` returns) and the
shift rule the reference configures (nn/qdense.py:246, 1400, 1596)."""
import pytest
import torch

from oracle import circuits as oc

pytestmark = pytest.mark.gpu


def _case(n, enc, imp, meas, L, S, batch, seed, feat=None, pad=0.0):
    from qiddm_amd.circuit import Circuit
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(1, L, S, n, 3, generator=g, dtype=torch.float64) * 0.8
    f = feat if feat is not None else n
    x = torch.rand(batch, f, generator=g, dtype=torch.float64) * 2 - 0.5
    if enc == "amplitude":
        x = x.abs() + 0.05
    circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=meas, n_blocks=L, sel_layers=S,
                   n_features=f if enc == "amplitude" else 0, pad_with=pad)
    spec = oc.Spec(n=n, encoding=enc, imprimitive=imp, measure=meas, pad_with=pad)
    cols = 2 ** n if meas == "probs" else n
    gout = torch.randn(batch, cols, generator=g, dtype=torch.float64)
    return circ, spec, x, w, gout


def _oracle_grads(spec, x, w, gout, wrt_x):
    w = w.clone().requires_grad_(True)
    x = x.clone().requires_grad_(wrt_x)
    loss = (oc.run_circuit(spec, x, w) * gout).sum()
    grads = torch.autograd.grad(loss, [w, x] if wrt_x else [w])
    return grads[0], (grads[1] if wrt_x else None)


@pytest.mark.parametrize("precision,tol", [("f64", dict(atol=1e-9, rtol=1e-9)), ("f32", dict(atol=2e-4, rtol=2e-3))])
@pytest.mark.parametrize("n,enc,imp,meas,L,S", [
    (1, "rz", "CZ", "expz", 2, 2),
    (3, "rz", "CZ", "expz", 2, 2),
    (4, "rz", "CZ", "probs", 3, 2),
    (6, "rz", "CNOT", "expz", 2, 3),
    (8, "rz", "CZ", "expz", 2, 2),
    (7, "ry", "CNOT", "probs", 1, 3),
    (5, "amplitude", "CNOT", "probs", 1, 3),
    (9, "rz", "CZ", "probs", 1, 2),
])
def test_shift_sweep_matches_autograd(n, enc, imp, meas, L, S, precision, tol):
    from qiddm_amd.circuit import run_shift_sweep
    feat = 20 if enc == "amplitude" else None
    circ, spec, x, w, gout = _case(n, enc, imp, meas, L, S, batch=13, seed=n * 31 + L, feat=feat, pad=0.2)
    wrt_x = enc in ("rz", "ry")
    ga, gi = run_shift_sweep(circ, x.cuda(), w.cuda(), gout.cuda(), precision, with_inputs=wrt_x)
    torch.cuda.synchronize()
    ra, ri = _oracle_grads(spec, x, w, gout, wrt_x)
    assert torch.allclose(ga.cpu(), ra, **tol), (ga.cpu() - ra).abs().max()
    if wrt_x:
        assert torch.allclose(gi.cpu(), ri[:, :n], **tol), (gi.cpu() - ri[:, :n]).abs().max()


def test_chunked_sweep_equals_single_sweep():
    from qiddm_amd.circuit import run_shift_sweep
    circ, spec, x, w, gout = _case(4, "rz", "CZ", "expz", 3, 2, batch=50, seed=9)
    a1, i1 = run_shift_sweep(circ, x.cuda(), w.cuda(), gout.cuda(), "f64")
    a2, i2 = run_shift_sweep(circ, x.cuda(), w.cuda(), gout.cuda(), "f64", max_dots_elems=50 * 10)
    assert torch.equal(a1, a2) and torch.equal(i1, i2)


def test_autograd_function_end_to_end():
    """execute(): two chained rounds, grads flow to the weights of both rounds and to
    the inputs (differN chaining, nn/qdense.py:464-465)."""
    from qiddm_amd.circuit import Circuit, execute
    n = 4
    g = torch.Generator().manual_seed(4)
    w = torch.randn(2, 2, 2, n, 3, generator=g, dtype=torch.float64) * 0.7
    x = torch.rand(6, n, generator=g, dtype=torch.float64)
    gout = torch.randn(6, 2 ** n, generator=g, dtype=torch.float64)
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="probs", n_rounds=2,
                   n_blocks=2, sel_layers=2)
    wd = w.cuda().requires_grad_(True)
    xd = x.cuda().requires_grad_(True)
    out = execute(circ, xd, wd, "f64")
    (out * gout.cuda()).sum().backward()
    spec = oc.Spec(n=n, encoding="rz", imprimitive="CZ", measure="probs")
    ra, ri = _oracle_grads(spec, x, w, gout, True)
    assert torch.allclose(out.detach().cpu(), oc.run_circuit(spec, x, w), atol=1e-11)
    assert torch.allclose(wd.grad.cpu(), ra, atol=1e-9)
    assert torch.allclose(xd.grad.cpu(), ri, atol=1e-9)
