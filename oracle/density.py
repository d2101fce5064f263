"""Oracle: density-matrix simulation with PennyLane's channel definitions (complex128, dense).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  **Parity unpinned** (PennyLane's ``default.mixed`` is not
installable here); the Kraus operators are the published ones [PL-0.29 ``qml.PhaseDamping`` /
``AmplitudeDamping`` / ``DepolarizingChannel`` docstrings]:
    PhaseDamping(g):      K0 = diag(1, sqrt(1-g)),  K1 = diag(0, sqrt(g))
    AmplitudeDamping(g):  K0 = diag(1, sqrt(1-g)),  K1 = [[0, sqrt(g)], [0, 0]]
    Depolarizing(p):      K0 = sqrt(1-p) I,  K1..3 = sqrt(p/3) X, Y, Z
rho is (B, D, D); wire w is bit n-1-w of both indices (wire 0 most significant), as in ``oracle.statevector``.
Call sites being restated: nn/qdense.py:95-106 (QDenseUndirected_old_noise), :249-265 (QNN_noise), :422-441
(differN_noise), :1403-1421 / :1599-1617 (QIDDM_PL/LL_noise).
"""
from __future__ import annotations

import math

import torch

from . import statevector as sv

CDT = torch.complex128


def from_state(psi, n):
    """(B, 2, ..., 2) or (B, D) state -> rho (B, D, D)."""
    v = psi.reshape(psi.shape[0], -1).to(CDT)
    return v.unsqueeze(2) * v.conj().unsqueeze(1)


def _apply_left(rho, mat, wire, n):
    """mat (2,2) or (B,2,2) on the row index."""
    b = rho.shape[0]
    d = 1 << n
    r = rho.reshape((b,) + (2,) * n + (d,))
    r = torch.movedim(r, 1 + wire, -2)                       # (..., 2, d)
    m = mat.to(CDT)
    if m.dim() == 2:
        r = torch.einsum("xy,...yd->...xd", m, r)
    else:
        shp = r.shape
        r = torch.einsum("bxy,bkyd->bkxd", m, r.reshape(b, -1, 2, d)).reshape(shp)
    r = torch.movedim(r, -2, 1 + wire)
    return r.reshape(b, d, d)


def apply_kraus(rho, kraus, wire, n):
    out = torch.zeros_like(rho)
    for k in kraus:
        left = _apply_left(rho, k, wire, n)
        # right-multiply by K^dagger == (K applied to the rows of the conjugate transpose)^dagger
        right = _apply_left(left.conj().transpose(1, 2), k, wire, n).conj().transpose(1, 2)
        out = out + right
    return out


def apply_unitary(rho, mat, wire, n):
    return apply_kraus(rho, [mat], wire, n)


def apply_diag_pair(rho, c, t, n, kind):
    """CZ / CNOT on (control c, target t): permutation / sign acting on both indices."""
    d = 1 << n
    idx = torch.arange(d)
    cb, tb = (idx >> (n - 1 - c)) & 1, (idx >> (n - 1 - t)) & 1
    if kind == "CZ":
        s = (1 - 2 * (cb & tb)).to(CDT)
        return rho * s.unsqueeze(0).unsqueeze(2) * s.unsqueeze(0).unsqueeze(1)
    perm = idx ^ (cb << (n - 1 - t))                          # new[perm[k]] = old[k]  (an involution)
    return rho[:, perm][:, :, perm]


def channel_kraus(name, p):
    z = torch.zeros(2, 2, dtype=CDT)
    if name == "PhaseDamping":
        k0, k1 = z.clone(), z.clone()
        k0[0, 0], k0[1, 1] = 1, math.sqrt(1 - p)
        k1[1, 1] = math.sqrt(p)
        return [k0, k1]
    if name == "AmplitudeDamping":
        k0, k1 = z.clone(), z.clone()
        k0[0, 0], k0[1, 1] = 1, math.sqrt(1 - p)
        k1[0, 1] = math.sqrt(p)
        return [k0, k1]
    if name == "DepolarizingChannel":
        eye = torch.eye(2, dtype=CDT)
        x = torch.tensor([[0, 1], [1, 0]], dtype=CDT)
        y = torch.tensor([[0, -1j], [1j, 0]], dtype=CDT)
        zz = torch.tensor([[1, 0], [0, -1]], dtype=CDT)
        return [math.sqrt(1 - p) * eye] + [math.sqrt(p / 3) * m for m in (x, y, zz)]
    raise ValueError(name)


def sel(rho, weights, n, imprimitive):
    w = weights.to(sv.RDT)
    for layer in range(w.shape[0]):
        for wire in range(n):
            rho = apply_unitary(rho, sv.rot_matrix(w[layer, wire, 0], w[layer, wire, 1], w[layer, wire, 2]), wire, n)
        if n > 1:
            r = layer % (n - 1) + 1
            for i in range(n):
                rho = apply_diag_pair(rho, i, (i + r) % n, n, imprimitive)
    return rho


def rz_batched(rho, angles, wire, n):
    a = angles.to(sv.RDT)
    m = torch.zeros(a.shape[0], 2, 2, dtype=CDT)
    m[:, 0, 0] = torch.exp(-0.5j * a)
    m[:, 1, 1] = torch.exp(0.5j * a)
    return apply_unitary(rho, m, wire, n)


def probs(rho):
    return torch.diagonal(rho, dim1=1, dim2=2).real


def expval_z(rho, n):
    p = probs(rho)
    idx = torch.arange(1 << n)
    return torch.stack([(p * (1 - 2 * ((idx >> (n - 1 - w)) & 1)).to(p.dtype)).sum(dim=1) for w in range(n)], dim=1)


def zero_rho(b, n):
    rho = torch.zeros(b, 1 << n, 1 << n, dtype=CDT)
    rho[:, 0, 0] = 1
    return rho
