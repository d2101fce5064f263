"""Density-matrix execution of a recorded quantum function: what ``qml.device("default.mixed", wires=n)`` runs
(reference: the noise study re-creates the layers' QNodes on it, src/mnist_noise.py:214-229, and the ``_circuit``
bodies insert PhaseDamping / AmplitudeDamping / DepolarizingChannel, nn/qdense.py:98-104, 255-261, 1410-1417).

The tape is lowered, in order, to the op program of ``qiddm_mixed_forward`` (templates and entangler rings expanded
here; see include/qiddm_hip.h) and executed in one launch, one workgroup per sample.  Forward only -- the reference
never differentiates on ``default.mixed`` -- and n <= 8.  No CPU path.
"""
from __future__ import annotations

import ctypes

import torch

from . import _capi

_workspaces = {}
CHANNELS = {"PhaseDamping": _capi.MIX_PHASE_DAMP, "AmplitudeDamping": _capi.MIX_AMP_DAMP,
            "DepolarizingChannel": _capi.MIX_DEPOL}


def rot_matrices(weights: torch.Tensor) -> torch.Tensor:
    """(..., 3) Rot angles -> (G, 8) float64 rows (u00, u01, u10, u11) as (re, im); Rot = RZ(omega) RY(theta) RZ(phi)."""
    w = weights.detach().to(torch.float64).reshape(-1, 3)
    phi, theta, omega = w[:, 0], w[:, 1], w[:, 2]
    c, s = torch.cos(theta / 2), torch.sin(theta / 2)
    a, b = (phi + omega) / 2, (phi - omega) / 2
    return torch.stack([torch.cos(a) * c, -torch.sin(a) * c, -torch.cos(b) * s, -torch.sin(b) * s,
                        torch.cos(b) * s, -torch.sin(b) * s, torch.cos(a) * c, torch.sin(a) * c], dim=1).contiguous()


class _Lowering:
    def __init__(self, n):
        self.n, self.ops, self.rows, self.gates = n, [], [], []
        self.batch, self.batched = None, False
        self.features, self.pad_with = None, 0.0
        self.device = None

    def _see(self, t):
        if torch.is_tensor(t):
            if not t.is_cuda:
                raise RuntimeError("default.mixed runs on the GPU only: tensors must live on a HIP device (no CPU path)")
            self.device = self.device or t.device

    def _note_batch(self, b, batched):
        if self.batch is None:
            self.batch, self.batched = b, batched
        elif self.batch != b:
            raise ValueError(f"inconsistent batch sizes in one circuit: {self.batch} vs {b}")

    def angle(self, value):
        """-> (row index or -1, constant)"""
        if not torch.is_tensor(value):
            return -1, float(value)
        self._see(value)
        if value.dim() == 0:
            self._note_batch(1, False)
            self.rows.append(value.detach().reshape(1))
        elif value.dim() == 1:
            self._note_batch(value.shape[0], True)
            self.rows.append(value.detach())
        else:
            raise NotImplementedError("gate parameters must be scalars or 1-D (batched) tensors")
        return len(self.rows) - 1, 0.0

    def op(self, kind, wire=0, a=-1, p=0.0, scale=1.0):
        self.ops.append((kind, wire, a, p, scale))

    def sel(self, weights, wires, imprimitive):
        n = len(wires)
        self._see(weights)
        base = sum(g.shape[0] for g in self.gates)
        self.gates.append(rot_matrices(weights))
        kind = _capi.MIX_CZ if imprimitive == "CZ" else _capi.MIX_CNOT
        for layer in range(weights.shape[0]):
            for i, w in enumerate(wires):
                self.op(_capi.MIX_GATE, w, base + layer * n + i)
            if n > 1:
                r = layer % (n - 1) + 1
                for i in range(n):
                    self.op(kind, wires[i], wires[(i + r) % n])


def lower(tape, ret, n):
    from . import qml
    low = _Lowering(n)
    all_w = tuple(range(n))
    first = True
    for t in tape:
        if t.name == "AmplitudeEmbedding":
            if not first or t.wires != all_w:
                raise NotImplementedError("AmplitudeEmbedding must come first and act on all wires")
            if not (t.hyper["normalize"] or t.hyper["pad_with"] is not None):
                raise NotImplementedError("AmplitudeEmbedding without normalize/pad_with")
            f = t.params[0]
            low._see(f)
            feat = f.shape[-1]
            if feat > (1 << n):
                raise ValueError(f"Features must be of length {1 << n} or smaller; got length {feat}.")
            if feat < (1 << n) and t.hyper["pad_with"] is None:
                raise ValueError(f"Features must be of length {1 << n}; got length {feat}. "
                                 "Use the 'pad_with' argument for automated padding.")
            low._note_batch(1 if f.dim() == 1 else f.shape[0], f.dim() > 1)
            low.features = f.detach().reshape(-1, feat)
            low.pad_with = float(t.hyper["pad_with"] or 0.0)
            low.op(_capi.MIX_AMP_EMBED)
        else:
            if first:
                low.op(_capi.MIX_ZERO)
            if t.name == "AngleEmbedding":
                if t.hyper["rotation"] != "Y":
                    raise NotImplementedError("only AngleEmbedding(rotation='Y') is supported")
                f = t.params[0]
                for i, w in enumerate(t.wires):
                    row, const = low.angle(f[..., i])
                    low.op(_capi.MIX_RY, w, row, const)
            elif t.name in ("RZ", "PhaseShift", "RY"):
                row, const = low.angle(t.params[0])
                low.op(_capi.MIX_RY if t.name == "RY" else _capi.MIX_PHASE, t.wires[0], row, const)
            elif t.name == "StronglyEntanglingLayers":
                low.sel(t.params[0], t.wires, t.hyper["imprimitive"])
            elif t.name in ("CZ", "CNOT"):
                low.op(_capi.MIX_CZ if t.name == "CZ" else _capi.MIX_CNOT, t.wires[0], t.wires[1])
            elif t.name in CHANNELS:
                p = t.params[0]
                low.op(CHANNELS[t.name], t.wires[0], -1, float(p))
            else:
                raise NotImplementedError(f"operation {t.name} is not supported on default.mixed")
        first = False
    if first:
        low.op(_capi.MIX_ZERO)
    # measurement
    if isinstance(ret, qml._Measurement):
        if ret.kind != "probs" or (ret.wires is not None and ret.wires != all_w):
            raise NotImplementedError("probs must cover wires 0..n-1")
        measure, as_list = _capi.MEAS_PROBS, False
    elif isinstance(ret, (list, tuple)) and all(isinstance(m, qml._Measurement) for m in ret) and \
            [m.kind for m in ret] == ["expz"] * n and [m.wires for m in ret] == [(i,) for i in range(n)]:
        measure, as_list = _capi.MEAS_EXPZ, True
    else:
        raise NotImplementedError("measurements must be probs(all wires) or [expval(PauliZ(i)) for i in range(n)]")
    return low, measure


def execute(tape, ret, n, precision=None):
    """Run the recorded function on the density-matrix kernel.  Returns float64 ``(B, 2^n)`` / ``(B, n)``
    (or the unbatched row), as ``default.mixed`` does."""
    from . import circuit as _c
    low, measure = lower(tape, ret, n)
    if low.device is None:
        raise RuntimeError("default.mixed runs on the GPU only: no tensor argument lives on a HIP device (no CPU path)")
    device = low.device
    batch = low.batch or 1
    prec = _capi.F64 if (precision or _c._default_precision) == "f64" else _capi.F32
    prog = (_capi.MixedOp * len(low.ops))()
    for dst, (kind, wire, a, p, scale) in zip(prog, low.ops):
        dst.kind, dst.wire, dst.a, dst.reserved, dst.p, dst.scale = kind, wire, a, 0, p, scale
    f64 = dict(dtype=torch.float64, device=device)
    rows = torch.stack([r.to(**f64).expand(batch) for r in low.rows]).contiguous() if low.rows else None
    gates = torch.cat(low.gates).to(device).contiguous() if low.gates else None
    feats = low.features.to(**f64).contiguous() if low.features is not None else None
    if feats is not None and feats.shape[0] != batch:
        feats = feats.expand(batch, -1).contiguous()
    out = torch.empty(batch, (1 << n) if measure == _capi.MEAS_PROBS else n, **f64)
    lib = _capi.lib()
    need = lib.qiddm_mixed_workspace_bytes(n, prec, batch, len(low.ops))
    if need < 0:
        _capi.check(int(need))
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < need:
        ws = _workspaces[key] = torch.empty(max(need, 256), dtype=torch.uint8, device=device)

    def ptr(t):
        return 0 if t is None else t.data_ptr()

    _capi.check(lib.qiddm_mixed_forward(
        n, prec, prog, len(low.ops), ptr(rows), 0 if rows is None else rows.stride(0), len(low.rows), ptr(feats),
        0 if feats is None else feats.stride(0), 0 if feats is None else feats.shape[1], 0.0, low.pad_with, ptr(gates),
        0 if gates is None else gates.shape[0], measure, batch, out.data_ptr(), out.stride(0), ws.data_ptr(), ws.numel(),
        ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
    return out if low.batched else out[0]
