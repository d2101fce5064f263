"""Oracle implementation (b): dense 2**n x 2**n unitaries from Kronecker products.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  **Parity unpinned.**

Deliberately shares no code with ``oracle.statevector``: plain numpy, every
gate is promoted to the full Hilbert space with ``np.kron`` and the
controlled gates are built from projectors, so a wire-order or sign slip in
one implementation shows up as a disagreement (known-answer test KA9).
Usable for n <= 8 or so.
"""
from __future__ import annotations

import numpy as np

I2 = np.eye(2, dtype=np.complex128)
X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
P0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)
P1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)


def rz(phi: float) -> np.ndarray:
    return np.array([[np.exp(-0.5j * phi), 0], [0, np.exp(0.5j * phi)]], dtype=np.complex128)


def ry(theta: float) -> np.ndarray:
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return np.array([[c, -s], [s, c]], dtype=np.complex128)


def rot(phi: float, theta: float, omega: float) -> np.ndarray:
    return rz(omega) @ ry(theta) @ rz(phi)


def _kron_all(mats) -> np.ndarray:
    out = np.array([[1.0 + 0j]])
    for m in mats:  # wire 0 first => most significant
        out = np.kron(out, m)
    return out


def lift_1q(u: np.ndarray, wire: int, n: int) -> np.ndarray:
    return _kron_all([u if w == wire else I2 for w in range(n)])


def lift_controlled(u: np.ndarray, c: int, t: int, n: int) -> np.ndarray:
    """|0><0|_c (x) I + |1><1|_c (x) u_t."""
    a = _kron_all([P0 if w == c else I2 for w in range(n)])
    b = _kron_all([P1 if w == c else (u if w == t else I2) for w in range(n)])
    return a + b


def sel_unitary(weights: np.ndarray, n: int, imprimitive: str = "CNOT") -> np.ndarray:
    """Full unitary of ``StronglyEntanglingLayers(weights[S,n,3])``."""
    weights = np.asarray(weights, dtype=np.float64)
    total = np.eye(2 ** n, dtype=np.complex128)
    two_q = X if imprimitive == "CNOT" else Z
    for l in range(weights.shape[0]):
        layer = _kron_all([rot(*weights[l, i]) for i in range(n)])
        total = layer @ total
        if n > 1:
            r = (l % (n - 1)) + 1
            for i in range(n):
                total = lift_controlled(two_q, i, (i + r) % n, n) @ total
    return total


def rz_layer_unitary(x: np.ndarray, n: int, scale: float = 1.0) -> np.ndarray:
    return _kron_all([rz(scale * x[j]) for j in range(n)])


def ry_layer_unitary(x: np.ndarray, n: int) -> np.ndarray:
    return _kron_all([ry(x[j]) for j in range(n)])


def amp_embed(x: np.ndarray, n: int, pad_with: float) -> np.ndarray:
    d = 2 ** n
    v = np.full(d, pad_with, dtype=np.float64)
    v[: len(x)] = x
    return (v / np.linalg.norm(v)).astype(np.complex128)


def probs(psi: np.ndarray) -> np.ndarray:
    return np.abs(psi) ** 2


def expval_z(psi: np.ndarray, n: int) -> np.ndarray:
    p = probs(psi)
    return np.array([np.real(np.vdot(psi, lift_1q(Z, i, n) @ psi)) for i in range(n)]) if n <= 6 else \
        np.array([np.sum(p * (1 - 2 * ((np.arange(2 ** n) >> (n - 1 - i)) & 1))) for i in range(n)])
