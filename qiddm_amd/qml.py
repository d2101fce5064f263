"""The slice of the PennyLane front-end the QIDDM layers use, bound to the HIP engine.

The reference builds its quantum layers as
``qml.QNode(func=self._circuit, device=qml.device(name, wires=n), interface="torch",
diff_method=...)`` and calls ``self.qnode(inputs[, weights])`` (reference
nn/qdense.py:26-38, 237-247, 406-420; nn/qconv.py:39-47).  This module keeps that
surface -- same names, same argument meaning, same errors for what a pure-state
device cannot do -- so that ``_circuit`` bodies written against PennyLane run
unchanged with ``import qiddm_amd.qml as qml``.

A QNode call records the quantum function into a tape, recognises the circuit
family of include/qiddm_hip.h (embedding | per-block RZ/RY encoders, SEL blocks,
probs | <Z_i>) and executes it in one launch of the fused HIP kernel.  Anything
outside that family raises ``NotImplementedError`` -- there is no gate-by-gate
or CPU fallback.
"""
from __future__ import annotations

import threading
from typing import Iterable

import torch

from . import circuit as _c

_tls = threading.local()

_PURE_STATE_DEVICES = ("default.qubit", "default.qubit.torch", "default.qubit.jax", "lightning.qubit",
                       "qiddm.hip")
_DIFF_METHODS = ("backprop", "parameter-shift", "adjoint", "best", None)


class DeviceError(Exception):
    """Same name PennyLane uses for unsupported operations on a device."""


class QuantumFunctionError(Exception):
    pass


# ---------------------------------------------------------------------------
# tape
# ---------------------------------------------------------------------------
class _Op:
    __slots__ = ("name", "wires", "params", "hyper")

    def __init__(self, name, wires, params=(), **hyper):
        self.name, self.wires, self.params, self.hyper = name, _wires(wires), tuple(params), hyper
        tape = getattr(_tls, "tape", None)
        if tape is not None:
            tape.append(self)

    def __repr__(self):
        return f"{self.name}(wires={list(self.wires)})"


def _wires(w) -> tuple:
    if isinstance(w, int):
        return (w,)
    if isinstance(w, Iterable):
        return tuple(int(i) for i in w)
    raise ValueError(f"bad wires {w!r}")


# --- operations the reference circuits use ----------------------------------
def RZ(phi, wires):
    return _Op("RZ", wires, (phi,))


def RY(phi, wires):
    return _Op("RY", wires, (phi,))


def PhaseShift(phi, wires):
    return _Op("PhaseShift", wires, (phi,))


def AmplitudeEmbedding(features, wires, pad_with=None, normalize=False):
    return _Op("AmplitudeEmbedding", wires, (features,), pad_with=pad_with, normalize=normalize)


def AngleEmbedding(features, wires, rotation="X"):
    return _Op("AngleEmbedding", wires, (features,), rotation=rotation)


class _Imprimitive:
    def __init__(self, name):
        self.name = name

    def __call__(self, wires):
        return _Op(self.name, wires)

    def __repr__(self):
        return self.name


CNOT = _Imprimitive("CNOT")
CZ = _Imprimitive("CZ")


class ops:  # ``qml.ops.CZ`` as spelled at reference nn/qdense.py:263
    CNOT = CNOT
    CZ = CZ


class StronglyEntanglingLayers:
    """``qml.StronglyEntanglingLayers(weights[S, n, 3], wires, ranges=None, imprimitive=CNOT)``."""

    def __new__(cls, weights, wires, ranges=None, imprimitive=None):
        if ranges is not None:
            raise NotImplementedError("custom StronglyEntanglingLayers ranges are not supported")
        imp = imprimitive if imprimitive is not None else CNOT
        w = _wires(wires)
        if weights.dim() != 3 or weights.shape[1] != len(w) or weights.shape[2] != 3:
            raise ValueError(f"Weights tensor must have shape (S, {len(w)}, 3); got {tuple(weights.shape)}")
        return _Op("StronglyEntanglingLayers", w, (weights,), imprimitive=getattr(imp, "name", str(imp)))

    @staticmethod
    def shape(n_layers, n_wires):
        return n_layers, n_wires, 3


def _channel(name):
    def make(p, wires):
        return _Op(name, wires, (p,), channel=True)
    make.__name__ = name
    return make


PhaseDamping = _channel("PhaseDamping")
AmplitudeDamping = _channel("AmplitudeDamping")
DepolarizingChannel = _channel("DepolarizingChannel")


# --- measurements -------------------------------------------------------------
class PauliZ:
    def __init__(self, wires):
        self.wires = _wires(wires)


class _Measurement:
    def __init__(self, kind, wires):
        self.kind, self.wires = kind, wires


def probs(wires=None):
    return _Measurement("probs", None if wires is None else _wires(wires))


def expval(obs):
    if not isinstance(obs, PauliZ):
        raise NotImplementedError("only expval(PauliZ(i)) is supported")
    return _Measurement("expz", obs.wires)


# ---------------------------------------------------------------------------
# device / QNode
# ---------------------------------------------------------------------------
class Device:
    def __init__(self, name, wires, **kwargs):
        if isinstance(wires, int):
            self.num_wires = wires
        else:
            self.num_wires = len(_wires(wires))
        self.short_name = name
        self.mixed = name == "default.mixed"
        if name not in _PURE_STATE_DEVICES and not self.mixed:
            raise DeviceError(f"Device {name} does not exist. Make sure the required plugin is installed.")

    def __repr__(self):
        kind = "density-matrix" if self.mixed else "statevector"
        return f"<qiddm_amd HIP {kind} device standing in for {self.short_name!r}, wires={self.num_wires}>"


def device(name, wires=1, **kwargs):
    """``qml.device(name, wires=n)``.  Every pure-state device name the reference uses maps
    to the HIP statevector engine; ``default.mixed`` (the noise scripts create it,
    src/mnist_noise.py:223) maps to the density-matrix kernel (``qiddm_amd.mixed``: forward only, n <= 8)."""
    return Device(name, wires, **kwargs)


class QNode:
    def __init__(self, func, device, interface="torch", diff_method="best", cache=True,
                 cachesize=10000, precision=None, **kwargs):
        if interface not in ("torch", "auto", None):
            raise QuantumFunctionError(f"Unknown interface {interface}. Interface must be 'torch'.")
        if diff_method not in _DIFF_METHODS:
            raise QuantumFunctionError(f"Differentiation method {diff_method} not recognized.")
        self.func = func
        self.device = device
        self.interface = interface
        self.diff_method = diff_method
        self.precision = precision
        self.circuit = None  # last compiled descriptor (introspection)

    # -- tracing ---------------------------------------------------------------
    def _trace(self, args, kwargs):
        if getattr(_tls, "tape", None) is not None:
            raise QuantumFunctionError("nested QNode calls are not supported")
        _tls.tape = []
        try:
            ret = self.func(*args, **kwargs)
            tape = _tls.tape
        finally:
            _tls.tape = None
        return tape, ret

    def __call__(self, *args, **kwargs):
        tape, ret = self._trace(args, kwargs)
        n = self.device.num_wires
        if self.device.mixed:
            # density-matrix execution (forward only; the reference samples, never trains, on default.mixed)
            from . import mixed as _mixed
            self.circuit = None
            with torch.no_grad():
                return _mixed.execute(tape, ret, n, self.precision)
        circ, x, angles, batched, as_list = _compile(tape, ret, n, self.device)
        self.circuit = circ
        out = _c.execute(circ, x, angles, self.precision, self.diff_method or "backprop")
        out_dtype = _result_dtype(x, angles)
        out = out.to(out_dtype)
        if not batched:
            out = out[0]
        return out


def _result_dtype(x, angles):
    # default.qubit.torch (R_DTYPE float64 / C_DTYPE complex128) and lightning.qubit both hand
    # back float64 whatever the parameter dtypes are (SURVEY.md finding F5)
    return torch.float64


# ---------------------------------------------------------------------------
# tape -> circuit descriptor
# ---------------------------------------------------------------------------
def _same_view(a, b) -> bool:
    return (torch.is_tensor(a) and torch.is_tensor(b) and a.data_ptr() == b.data_ptr()
            and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype)


def _compile(tape, ret, n, dev):
    for op in tape:
        if op.hyper.get("channel"):
            raise DeviceError(f"Gate {op.name} not supported on device {dev.short_name}")

    # measurement
    if isinstance(ret, _Measurement):
        meas, as_list = ret, False
        if meas.kind != "probs" or (meas.wires is not None and meas.wires != tuple(range(n))):
            if meas.kind == "expz" and n == 1 and meas.wires == (0,):
                pass
            else:
                raise NotImplementedError("probs must cover wires 0..n-1")
        measure = meas.kind
    elif isinstance(ret, (list, tuple)) and all(isinstance(m, _Measurement) for m in ret):
        if [m.kind for m in ret] != ["expz"] * n or [m.wires for m in ret] != [(i,) for i in range(n)]:
            raise NotImplementedError("expectation values must be [expval(PauliZ(i)) for i in range(n)]")
        measure, as_list = "expz", True
    else:
        raise QuantumFunctionError("A quantum function must return measurements")

    i = 0
    enc, x = "none", None
    pad, scale = 0.0, 1.0
    all_w = tuple(range(n))
    if i < len(tape) and tape[i].name == "AmplitudeEmbedding":
        op = tape[i]
        if op.wires != all_w:
            raise NotImplementedError("AmplitudeEmbedding must act on all wires")
        if not (op.hyper["normalize"] or op.hyper["pad_with"] is not None):
            raise NotImplementedError("AmplitudeEmbedding without normalize/pad_with")
        enc, x = "amplitude", op.params[0]
        feat = x.shape[-1]
        if feat < (1 << n) and op.hyper["pad_with"] is None:
            raise ValueError(f"Features must be of length {1 << n}; got length {feat}. "
                             "Use the 'pad_with' argument for automated padding.")
        pad = float(op.hyper["pad_with"] or 0.0)
        i += 1
    elif i < len(tape) and tape[i].name == "AngleEmbedding":
        op = tape[i]
        if op.hyper["rotation"] != "Y" or op.wires != all_w:
            raise NotImplementedError("only AngleEmbedding(rotation='Y') on all wires is supported")
        enc, x = "ry", op.params[0]
        if x.shape[-1] != n:
            raise ValueError(f"Features must be of length {n}; got length {x.shape[-1]}.")
        i += 1

    blocks, rz_cols, imp = [], None, None
    while i < len(tape):
        op = tape[i]
        if op.name in ("RZ", "RY"):
            kind, this_enc = op.name, ("rz" if op.name == "RZ" else "ry_blocks")
            cols = []
            for j in range(n):
                if i >= len(tape) or tape[i].name != kind or tape[i].wires != (j,):
                    raise NotImplementedError(f"{kind} encoders must cover wires 0..n-1 in order")
                cols.append(tape[i].params[0])
                i += 1
            if enc not in ("none", this_enc) or (enc == "none" and blocks):
                raise NotImplementedError("mixed encodings in one circuit")
            if rz_cols is None:
                rz_cols = cols
            elif not all(_same_view(a, b) for a, b in zip(rz_cols, cols)):
                raise NotImplementedError("every block must re-upload the same inputs")
            enc = this_enc
            if i >= len(tape) or tape[i].name != "StronglyEntanglingLayers":
                raise NotImplementedError("an encoder layer must be followed by StronglyEntanglingLayers")
            continue
        if op.name == "StronglyEntanglingLayers":
            if op.wires != all_w:
                raise NotImplementedError("StronglyEntanglingLayers must act on all wires")
            if enc in ("rz", "ry_blocks") and len(blocks) >= 1 and rz_cols is None:
                raise NotImplementedError("blocks without encoder after blocks with encoder")
            this_imp = op.hyper["imprimitive"]
            if imp is not None and this_imp != imp:
                raise NotImplementedError("mixed imprimitives")
            imp = this_imp
            blocks.append(op.params[0])
            i += 1
            continue
        if op.name == "PhaseShift":
            # diagonal gates directly in front of a computational-basis measurement do not
            # change probs / <Z>; they must be trailing
            if any(t.name != "PhaseShift" for t in tape[i:]):
                raise NotImplementedError("PhaseShift is only supported directly before the measurement")
            break
        raise NotImplementedError(f"operation {op.name} is outside the supported circuit family")
    if not blocks:
        raise NotImplementedError("circuit has no StronglyEntanglingLayers block")
    if enc in ("rz", "ry_blocks") and len(blocks) > 1:
        # the pattern requires an encoder in front of every block
        n_rz = sum(1 for t in tape if t.name == ("RZ" if enc == "rz" else "RY"))
        if n_rz != n * len(blocks):
            raise NotImplementedError("every StronglyEntanglingLayers block needs its RZ encoder layer")
    s_layers = blocks[0].shape[0]
    if any(b.shape != blocks[0].shape for b in blocks):
        raise NotImplementedError("all SEL blocks must have the same number of layers")
    angles = torch.stack([b for b in blocks], dim=0).unsqueeze(0)  # (1, L, S, n, 3)

    batched = True
    if enc in ("rz", "ry_blocks"):
        cols = [c if torch.is_tensor(c) else torch.as_tensor(c) for c in rz_cols]
        batched = cols[0].dim() >= 1
        x = torch.stack([c.reshape(-1) for c in cols], dim=-1)  # (B, n)
    elif enc in ("amplitude", "ry"):
        batched = x.dim() >= 2
        if not batched:
            x = x.unsqueeze(0)
    else:
        raise NotImplementedError("circuits without data encoding need an explicit batch; "
                                  "use qiddm_amd.circuit.run_forward")
    x = x.to(angles.device)
    circ = _c.Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=measure, n_rounds=1,
                      n_blocks=len(blocks), sel_layers=s_layers,
                      n_features=x.shape[-1] if enc == "amplitude" else 0, enc_scale=scale,
                      enc_offset=0.0, pad_with=pad)
    return circ, x, angles, batched, as_list
