"""Oracle implementation (a): gate-by-gate strided statevector update.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Pinning: the RZ /
Rot / SEL-range / CZ / <Z> / wire-order conventions below are **pinned** to
PennyLane-Lightning outputs the reference ships
(``tests/test_oracle_reference_runs.py``); ``amplitude_embedding``,
``apply_cnot`` and ``probs`` as a returned quantity are **parity unpinned**
(published definitions + known-answer tests; PennyLane is not available
offline).

State layout: ``(B, 2**n)`` complex128 torch tensor.  Wire ``w`` is bit
``n-1-w`` of the amplitude index (wire 0 = most significant), which is the
order ``qml.probs(wires=range(n))`` reports (reference call sites:
``nn/qdense.py:47,105,441``; ``nn/qconv.py:56``).

Everything here is written with differentiable torch ops so that
``torch.autograd`` through the oracle is the gradient reference for the HIP
parameter-shift / adjoint backward (known-answer test KA10).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch

CDT = torch.complex128
RDT = torch.float64


# ---------------------------------------------------------------------------
# elementary matrices
# ---------------------------------------------------------------------------
def _as_real(x) -> torch.Tensor:
    return torch.as_tensor(x, dtype=RDT) if not torch.is_tensor(x) else x.to(RDT)


def rz_diag(phi) -> torch.Tensor:
    """RZ(phi) = diag(exp(-i phi/2), exp(+i phi/2)).  Returns (..., 2)."""
    phi = _as_real(phi)
    half = 0.5 * phi
    e = torch.complex(torch.cos(half), torch.sin(half))
    return torch.stack([e.conj(), e], dim=-1)


def ry_matrix(theta) -> torch.Tensor:
    """RY(theta) = [[c, -s], [s, c]], c = cos(theta/2), s = sin(theta/2)."""
    theta = _as_real(theta)
    c = torch.cos(0.5 * theta)
    s = torch.sin(0.5 * theta)
    row0 = torch.stack([c, -s], dim=-1)
    row1 = torch.stack([s, c], dim=-1)
    return torch.stack([row0, row1], dim=-2).to(CDT)


def rot_matrix(phi, theta, omega) -> torch.Tensor:
    """Rot(phi, theta, omega) = RZ(omega) RY(theta) RZ(phi)  (RZ(phi) acts first).

    This is the single-qubit gate ``StronglyEntanglingLayers`` applies with
    ``weights[l, i, :] = (phi, theta, omega)`` (K4 in SURVEY.md section 2).
    """
    dz_phi = rz_diag(phi)
    dz_om = rz_diag(omega)
    ry = ry_matrix(theta)
    # RZ(om) @ RY @ RZ(phi): rows scaled by dz_om, columns by dz_phi
    return dz_om.unsqueeze(-1) * ry * dz_phi.unsqueeze(-2)


# ---------------------------------------------------------------------------
# state helpers
# ---------------------------------------------------------------------------
def zero_state(batch: int, n: int) -> torch.Tensor:
    st = torch.zeros(batch, 2 ** n, dtype=CDT)
    st[:, 0] = 1.0
    return st


def apply_1q(state: torch.Tensor, u: torch.Tensor, wire: int, n: int) -> torch.Tensor:
    """Apply a 2x2 matrix ``u`` ((2,2) shared, or (B,2,2) per sample) on ``wire``."""
    b = state.shape[0]
    left = 2 ** wire
    right = 2 ** (n - wire - 1)
    st = state.reshape(b, left, 2, right)
    if u.dim() == 2:
        out = torch.einsum("ij,bljr->blir", u.to(CDT), st)
    else:
        out = torch.einsum("bij,bljr->blir", u.to(CDT), st)
    return out.reshape(b, 2 ** n)


def apply_diag_1q(state: torch.Tensor, d: torch.Tensor, wire: int, n: int) -> torch.Tensor:
    """Apply a diagonal gate, ``d`` of shape (2,) or (B,2), on ``wire``."""
    b = state.shape[0]
    left = 2 ** wire
    right = 2 ** (n - wire - 1)
    st = state.reshape(b, left, 2, right)
    if d.dim() == 1:
        out = st * d.to(CDT).reshape(1, 1, 2, 1)
    else:
        out = st * d.to(CDT).reshape(b, 1, 2, 1)
    return out.reshape(b, 2 ** n)


def _bit(index: torch.Tensor, wire: int, n: int) -> torch.Tensor:
    return (index >> (n - 1 - wire)) & 1


def apply_cz(state: torch.Tensor, c: int, t: int, n: int) -> torch.Tensor:
    """CZ: sign flip on basis states with both wires set (K6)."""
    k = torch.arange(2 ** n)
    sign = 1.0 - 2.0 * (_bit(k, c, n) & _bit(k, t, n)).to(RDT)
    return state * sign.to(CDT)


def apply_cnot(state: torch.Tensor, c: int, t: int, n: int) -> torch.Tensor:
    """CNOT(control c, target t): amplitude permutation (K5).

    new[k] = old[k ^ (bit_c(k) << pos_t)]  (the map is an involution).
    """
    k = torch.arange(2 ** n)
    src = k ^ (_bit(k, c, n) << (n - 1 - t))
    return state[:, src]


# ---------------------------------------------------------------------------
# templates
# ---------------------------------------------------------------------------
def sel_ranges(n_layers: int, n: int) -> list:
    """Default ``StronglyEntanglingLayers`` ranges: r_l = (l mod (n-1)) + 1."""
    if n <= 1:
        return [0] * n_layers
    return [(l % (n - 1)) + 1 for l in range(n_layers)]


def strongly_entangling_layers(state, weights, n: int, imprimitive: str = "CNOT"):
    """``qml.StronglyEntanglingLayers(weights[S, n, 3], wires=range(n), imprimitive=...)``.

    Per layer: Rot on every wire, then (if n > 1) for i in 0..n-1 the
    imprimitive on (i, (i + r_l) mod n), control = first wire.
    """
    weights = _as_real(weights)
    s_layers = weights.shape[0]
    assert weights.shape[1:] == (n, 3), weights.shape
    ranges = sel_ranges(s_layers, n)
    for l in range(s_layers):
        for i in range(n):
            u = rot_matrix(weights[l, i, 0], weights[l, i, 1], weights[l, i, 2])
            state = apply_1q(state, u, i, n)
        if n > 1:
            for i in range(n):
                j = (i + ranges[l]) % n
                if imprimitive == "CNOT":
                    state = apply_cnot(state, i, j, n)
                elif imprimitive == "CZ":
                    state = apply_cz(state, i, j, n)
                else:
                    raise ValueError(imprimitive)
    return state


def amplitude_embedding(features, n: int, pad_with=None, normalize: bool = True):
    """``qml.AmplitudeEmbedding(features, wires=range(n), normalize, pad_with)`` (K1).

    Right-pad each row to 2**n with the constant, then divide by the L2 norm
    unless the norm already equals 1 within 1e-10 (PennyLane's TOLERANCE).
    """
    x = _as_real(features)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    b, f = x.shape
    d = 2 ** n
    if f > d:
        raise ValueError(f"Features must be of length {d} or smaller; got length {f}.")
    if f < d:
        if pad_with is None:
            raise ValueError(f"Features must be of length {d}; got length {f}. "
                             "Use the 'pad_with' argument for automated padding.")
        pad = torch.full((b, d - f), float(pad_with), dtype=RDT)
        x = torch.cat([x, pad], dim=1)
    norm2 = (x * x).sum(dim=1, keepdim=True)
    needs = (norm2 - 1.0).abs() > 1e-10
    if bool(needs.any()):
        if not (normalize or pad_with is not None):
            raise ValueError("Features must be a vector of norm 1.0; use 'normalize=True'.")
        scale = torch.where(needs, torch.sqrt(norm2), torch.ones_like(norm2))
        x = x / scale
    return x.to(CDT)


def angle_embedding_rz(state, inputs, n: int, scale: float = 1.0):
    """``for j in range(n): qml.RZ(inputs[:, j], wires=j)`` (K2, per-sample angles)."""
    x = _as_real(inputs)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    for j in range(n):
        state = apply_diag_1q(state, rz_diag(scale * x[:, j]), j, n)
    return state


def angle_embedding_ry(state, inputs, n: int):
    """``qml.AngleEmbedding(features, wires=range(n), rotation="Y")`` (K3)."""
    x = _as_real(inputs)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    for j in range(min(n, x.shape[1])):
        state = apply_1q(state, ry_matrix(x[:, j]), j, n)
    return state


def probs(state) -> torch.Tensor:
    """``qml.probs(wires=range(n))`` (K7)."""
    return state.real ** 2 + state.imag ** 2


def expval_z(state, n: int) -> torch.Tensor:
    """``[qml.expval(qml.PauliZ(i)) for i in range(n)]`` -> (B, n) (K8)."""
    p = probs(state)
    k = torch.arange(2 ** n)
    cols = []
    for i in range(n):
        sign = 1.0 - 2.0 * _bit(k, i, n).to(RDT)
        cols.append((p * sign).sum(dim=1))
    return torch.stack(cols, dim=1)
