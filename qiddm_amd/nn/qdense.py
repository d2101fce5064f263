"""Dense / "implicit" quantum denoisers with the reference's constructor
signatures, parameter names, ``save_name()`` strings and ``forward`` contracts
(reference nn/qdense.py), executing on the HIP statevector engine.

Each class cites the reference lines it mirrors.  Differences that are
deliberate (SURVEY.md "Findings"):

* F1 -- the ``lightning.qubit`` classes wrap every QNode result in
  ``torch.tensor(...)``, which detaches it, so their quantum weights and
  ``linear_down`` never receive gradients.  ``detach_quantum=True`` (default)
  reproduces that as-written behaviour; ``detach_quantum=False`` lets the
  parameter-shift gradient flow.
* the per-sample Python loops (``for i in range(b)``) are replaced by one batched
  launch -- same numbers, one wavefront per sample.
* F4 -- PCA front-ends stay host-side (sklearn), re-fit on every call as written.

There is no CPU execution path: ``forward`` raises unless the module lives on a
HIP device and the in-tree extension is built.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from .. import _capi
from .. import circuit as _c
from .. import pca as _pca
from .. import qml


def _pair(shape):
    return (shape, shape) if isinstance(shape, int) else tuple(shape)


def _qw_tanh(w):
    """qW-Map 0.1.2 ``qw_map.tanh`` (reference nn/qdense.py:45): pi * tanh(w)."""
    return math.pi * torch.tanh(w)


class _QuantumNet(nn.Module):
    """Shared plumbing: QNode construction, checkpoint helpers."""

    def _make_qnode(self, device_type: str, diff_method: str):
        self.device_type = device_type
        self.diff_method = diff_method
        self.qdev = qml.device(device_type, wires=self._n_wires())
        self.qnode = qml.QNode(func=self._circuit, device=self.qdev, interface="torch",
                               diff_method=diff_method)
        self._own_qnode = self.qnode

    def _n_wires(self) -> int:
        return getattr(self, "wires", None) or self.hidden_features

    def _add_noise_ops(self, wire, table):
        """Hardware-noise ops the reference inserts when add_noise != 0; on a pure-state
        device the channels raise DeviceError exactly as PennyLane does."""
        kind = getattr(self, "add_noise", 0)
        if kind in table:
            op, p = table[kind]
            op(p, wires=wire)

    # checkpoint helpers present on several reference classes (nn/qdense.py:297-307)
    def save_model(self, path, loss_values, epochs):
        torch.save({"model_state_dict": self.state_dict(), "loss_values": loss_values, "epochs": epochs}, path)

    def load_model(self, path):
        checkpoint = torch.load(path, map_location="cpu")
        self.load_state_dict(checkpoint["model_state_dict"])

    _fusable_noise = (0,)   # add_noise settings that leave the circuit a pure-state no-op

    def _fused_rounds_ok(self) -> bool:
        return (not torch.is_grad_enabled()) and self.qnode is self._own_qnode and \
            getattr(self, "add_noise", 0) in self._fusable_noise

    def _sampler_tables(self, circ, angles):
        """Per-layer tables of the fused sampler, rebuilt only when the weights (or the precision) changed."""
        stamp = (angles._version, angles.data_ptr(), str(angles.device), _c._default_precision, circ.angles_shape)
        cached = getattr(self, "_sampler_tables_cache", None)
        if cached is None or cached[0] != stamp:
            cached = (stamp, _c.dense_sample_tables(circ, angles.reshape(circ.angles_shape)))
            self._sampler_tables_cache = cached
        return cached[1]

    def _gate_table(self, circ, weights, angle_map=None):
        """Gate table of ``circ`` for the current weights (``qiddm_prepare_gates``), rebuilt only when they (or the
        precision) changed; ``angle_map`` is applied to the weights first (``qw_map.tanh`` / ``torch.tanh``)."""
        stamp = (weights._version, weights.data_ptr(), str(weights.device), _c._default_precision, circ.angles_shape,
                 circ.n_features)
        cached = getattr(self, "_gate_table_cache", None)
        if cached is None or cached[0] != stamp:
            ang = weights.detach() if angle_map is None else angle_map(weights.detach())
            ang = ang.reshape(circ.angles_shape)
            cached = (stamp, ang, _c.prepare_gates(circ, ang, _c._default_precision))
            self._gate_table_cache = cached
        return cached[1], cached[2]

    def _lean_sampler_tables(self, circ, angles, lin_down, lin_up):
        """Tables of the lean 8- / 6-qubit sampler, rebuilt only when the weights, the two linears or the
        precision changed; None when that kernel does not apply (then the general sampler runs)."""
        tensors = (angles, lin_down.weight, lin_down.bias, lin_up.weight, lin_up.bias)
        stamp = tuple((t._version, t.data_ptr()) for t in tensors if t is not None) + \
            (str(angles.device), _c._default_precision, circ.angles_shape)
        cached = getattr(self, "_lean_tables_cache", None)
        if cached is None or cached[0] != stamp:
            if torch.cuda.is_current_stream_capturing():
                return None                       # cannot validate inside a recording: not cached either
            cached = (stamp, _c.dense_sample_lean_tables(circ, angles.reshape(circ.angles_shape), lin_down.weight,
                                                         lin_down.bias, lin_up.weight, lin_up.bias))
            self._lean_tables_cache = cached
        return cached[1]

    def _fused_sampler_launch(self, circ, flat, angles, n_steps, goal, noise_factor):
        """n_steps loop bodies in one launch: the lean 8- / 6-qubit kernel where it applies, else the general
        four-wavefront sampler."""
        ld, lu = self.linear_down, self.linear_up
        post_mode = 0 if goal == "data" else 1
        lean = self._lean_sampler_tables(circ, angles, ld, lu)
        if lean is not None:
            return _c.dense_sample_lean(circ, flat, ld.weight, ld.bias, lu.weight, lu.bias, n_steps, lean,
                                        post_mode=post_mode, noise_factor=noise_factor)
        return _c.dense_sample(circ, flat, ld.weight, ld.bias, angles.reshape(circ.angles_shape), lu.weight, lu.bias,
                               n_steps, post_mode=post_mode, noise_factor=noise_factor,
                               tables=self._sampler_tables(circ, angles))

    # -- fused training step -------------------------------------------------
    def _train_family(self):
        """``(circuit, linear_down, angles parameter, linear_up)`` for nets of the
        linear_down -> angle-encoded circuit -> <Z> -> linear_up shape; None otherwise."""
        return None

    def fused_train_step(self, x, noise, schedule, goal, want_recon=False, want_elem_loss=False, rng_state=None):
        """What ``Diffusion.run_training_step_*`` does around this net -- noising, forward, MSE, backward --
        in three launches (``qiddm_train_step``).  Adds the gradients to ``.grad`` exactly where ``.backward()``
        would (with ``detach_quantum`` only ``linear_up`` receives one, finding F1) and returns the dict of
        ``circuit.train_step`` (``loss``, optional ``recon`` / ``elem_loss``); None when the net or its
        current settings are outside the fused step (the caller then runs the eager path)."""
        fam = self._train_family()
        if fam is None or self.qnode is not self._own_qnode or \
                getattr(self, "add_noise", 0) not in self._fusable_noise:
            return None
        circ, lin_down, angles, lin_up = fam
        if circ.n_qubits > 10 or not x.is_cuda or lin_up.weight.shape[0] != x.shape[1]:
            return None
        quantum = not self.detach_quantum
        res = _c.train_step(circ, x, noise, schedule, goal, lin_down.weight, lin_down.bias,
                            angles.reshape(circ.angles_shape), lin_up.weight, lin_up.bias, quantum,
                            want_recon=want_recon, want_elem_loss=want_elem_loss, rng_state=rng_state)
        pairs = [(lin_up.weight, res["w_up"]), (lin_up.bias, res["b_up"])]
        if quantum:
            pairs += [(lin_down.weight, res["w_down"]), (lin_down.bias, res["b_down"]), (angles, res["angles"])]
        for prm, g in pairs:
            if prm is None or not prm.requires_grad:
                continue
            g = g.to(prm.dtype).view_as(prm)
            if prm.grad is None:
                prm.grad = g
            else:
                prm.grad.add_(g)        # in place, as autograd accumulates: .grad may alias a flat DP bucket
        return res


# ===========================================================================
# A4: amplitude embedding + SEL(CNOT) + probs
# ===========================================================================
# QIDDM_NO_DENSE_UNITARY=1: always simulate every sample (A/B of the unitary route below)
_DENSE_UNITARY = os.environ.get("QIDDM_NO_DENSE_UNITARY") is None
_UNITARY_ROUTE_MIN_BATCH = 64      # (20 us at batch 256 against 171 us of simulation; the unitary itself costs one 2^n-sample
                                   #  simulation per weights)
class QDenseUndirected_old(_QuantumNet):
    """Reference nn/qdense.py:15-68.  ``(qdepth, shape)``; weights mapped with ``qw_map.tanh``."""

    _weight_map = staticmethod(_qw_tanh)

    def __init__(self, qdepth, shape) -> None:
        super().__init__()
        self._init_common(qdepth, shape)
        self._make_qnode("default.qubit.torch", "backprop")

    def _init_common(self, qdepth, shape):
        self.qdepth = qdepth
        self.width, self.height = _pair(shape)
        self.pixels = self.width * self.height
        self.wires = math.ceil(math.log2(self.pixels))
        weight_shape = qml.StronglyEntanglingLayers.shape(self.qdepth, self.wires)
        self.weights = nn.Parameter(torch.randn(weight_shape, requires_grad=True) * 0.4)

    def _circuit(self, inp):
        qml.AmplitudeEmbedding(features=inp, wires=range(self.wires), normalize=True, pad_with=0.1)
        qml.StronglyEntanglingLayers(weights=self._weight_map(self.weights), wires=range(self.wires))
        return qml.probs(wires=range(self.wires))

    def _post_process(self, probs):
        return torch.clamp(probs[:, : self.pixels] * self.pixels, 0, 1)

    def _unitary_operand(self, build=True):
        """``[Re U^T | Im U^T]`` of the weight-only layers for the first ``pixels`` outcomes, float32, rebuilt when the
        weights changed (``circuit.dense_unitary_forward``).  ``build=False``: only what is cached for the current
        weights, else None."""
        w = self.weights
        stamp = (w._version, w.data_ptr(), str(w.device))
        cached = getattr(self, "_unitary_operand_cache", None)
        if cached is None or cached[0] != stamp:
            if not build or torch.cuda.is_current_stream_capturing():
                return None                   # (a recording computes nothing: warm the cache with one eager call first)
            u = _c.circuit_unitary(self._weight_map(w.detach().double()), self.wires, "CNOT", precision="f64")
            cached = (stamp, _c.dense_unitary_operand(u, self.pixels))
            self._unitary_operand_cache = cached
        return cached[1]

    def forward(self, x):
        b = x.shape[0]
        flat = x.reshape(b, self.pixels)                     # "b 1 w h -> b (w h)"
        if self._fused_rounds_ok() and self.wires <= 10 and flat.is_cuda and \
                type(self)._circuit in (QDenseUndirected_old._circuit, QDenseUndirected_old_noise._circuit) and \
                type(self)._post_process is QDenseUndirected_old._post_process:
            # inference: embedding, circuit and post-processing in one launch (no (B, 2^n) probability matrix)
            circ = _c.Circuit(n_qubits=self.wires, encoding="amplitude", imprimitive="CNOT", measure="probs",
                              n_rounds=1, n_blocks=1, sel_layers=self.qdepth, n_features=self.pixels, pad_with=0.1)
            # (building the unitary costs one 2^n-sample simulation: worth it from _UNITARY_ROUTE_MIN_BATCH samples on;
            #  once it is there for the current weights every batch size takes it)
            operand = self._unitary_operand(build=b >= _UNITARY_ROUTE_MIN_BATCH) \
                if (_c._default_precision == "f32" and _DENSE_UNITARY) else None
            if operand is not None:
                # the circuit does not depend on the data: one float32 product with the circuit unitary (cached per
                # weights) instead of simulating every sample (C3's batch of 1024: 193 -> 45 us)
                out = _c.dense_unitary_forward(flat, operand, self.wires, self.pixels, 0.1, float(self.pixels))
                return out.reshape(b, 1, self.width, self.height)
            angles, table = self._gate_table(circ, self.weights, self._weight_map)
            out = _c.run_forward_post(circ, flat, angles, self.pixels, float(self.pixels), table=table)
            return out.reshape(b, 1, self.width, self.height)
        out = self._post_process(self.qnode(flat))
        return out.reshape(b, 1, self.width, self.height)

    def __repr__(self):
        return f"QDenseUndirected_old(qdepth={self.qdepth}, wires={self.wires})"

    def save_name(self) -> str:
        return f"QDenseUndirected_old{self.qdepth}_w{self.width}_h{self.height}"


class QDenseUndirected_old_noise(QDenseUndirected_old):
    """Reference nn/qdense.py:71-125.  ``(qdepth, shape, add_noise=0, device_type)``;
    weights mapped with plain ``torch.tanh`` (:97)."""

    _weight_map = staticmethod(torch.tanh)

    def __init__(self, qdepth, shape, add_noise=0, device_type="default.qubit.torch") -> None:
        nn.Module.__init__(self)
        self.add_noise = add_noise
        self._init_common(qdepth, shape)
        self._make_qnode(device_type, "backprop")

    def _circuit(self, inp):
        qml.AmplitudeEmbedding(features=inp, wires=range(self.wires), normalize=True, pad_with=0.1)
        qml.StronglyEntanglingLayers(weights=self._weight_map(self.weights), wires=range(self.wires))
        for wire in range(self.wires):
            self._add_noise_ops(wire, {1: (qml.PhaseShift, 0.05), 2: (qml.AmplitudeDamping, 0.1),
                                       3: (qml.DepolarizingChannel, 0.02)})
        return qml.probs(wires=range(self.wires))

    def __repr__(self):
        return f"QDenseUndirected_old_noise(qdepth={self.qdepth}, wires={self.wires}, add_noise={self.add_noise})"

    def save_name(self) -> str:
        return f"QDenseUndirected_old_noise{self.qdepth}_w{self.width}_h{self.height}_noise{self.add_noise}"


# ===========================================================================
# QNN_A: linear_down + AngleEmbedding(Y) + SEL(CNOT) + probs
# ===========================================================================
class QNN_A(_QuantumNet):
    """Reference nn/qdense.py:128-210."""

    def __init__(self, qdepth, shape, add_noise=0, device_type="default.qubit.torch",
                 diff_method="backprop") -> None:
        super().__init__()
        self.qdepth = qdepth
        self.add_noise = add_noise
        self.width, self.height = _pair(shape)
        self.pixels = self.width * self.height
        self.wires = math.ceil(math.log2(self.pixels))
        self.linear_down = nn.Linear(self.pixels, self.wires, dtype=torch.double)
        weight_shape = qml.StronglyEntanglingLayers.shape(self.qdepth, self.wires)
        self.weights = nn.Parameter(torch.randn(weight_shape, dtype=torch.double) * 0.4, requires_grad=True)
        self._make_qnode(device_type, diff_method)

    def _circuit(self, inp):
        qml.AngleEmbedding(features=inp, wires=range(self.wires), rotation="Y")
        qml.StronglyEntanglingLayers(weights=self.weights, wires=range(self.wires))
        for wire in range(self.wires):
            self._add_noise_ops(wire, {1: (qml.PhaseDamping, 0.05), 2: (qml.AmplitudeDamping, 0.05),
                                       3: (qml.DepolarizingChannel, 0.02)})
        return qml.probs(wires=range(self.wires))

    def _post_process(self, probs):
        return torch.clamp(probs[:, : self.pixels] * self.pixels, 0, 1)

    def forward(self, x):
        b = x.shape[0]
        red = self.linear_down(x.reshape(b, self.pixels))
        out = self._post_process(self.qnode(red))
        return out.reshape(b, 1, self.width, self.height)

    def __repr__(self):
        return f"QNN_A(qdepth={self.qdepth}, wires={self.wires}, add_noise={self.add_noise})"

    def save_name(self) -> str:
        return f"QNN_A{self.qdepth}_w{self.width}_h{self.height}_noise{self.add_noise}"


# ===========================================================================
# A1: linear_down -> RZ + SEL(CZ) + <Z> -> linear_up   (single SEL call)
# ===========================================================================
class QNN_noise(_QuantumNet):
    """Reference nn/qdense.py:219-307.  ``(input_dim, hidden_features, qdepth, add_noise=0)``."""

    def __init__(self, input_dim, hidden_features, qdepth: int, add_noise=0, detach_quantum=True) -> None:
        super().__init__()
        if isinstance(input_dim, str):
            input_dim = eval(input_dim)  # e.g. "28 * 28", as the reference accepts (:222-223)
        self.hidden_features = hidden_features
        self.qdepth = qdepth
        self.add_noise = add_noise
        self.detach_quantum = detach_quantum
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.linear_down = nn.Linear(input_dim, hidden_features, dtype=torch.double).to(self.device)
        self.linear_up = nn.Linear(hidden_features, input_dim, dtype=torch.double).to(self.device)
        weight_shape = qml.StronglyEntanglingLayers.shape(self.qdepth, self.hidden_features)
        self.weights = nn.Parameter(
            torch.randn(weight_shape, dtype=torch.double, requires_grad=True).to(self.device) * 0.4)
        self._make_qnode("lightning.qubit", "parameter-shift")

    _noise_table = {1: (qml.PhaseDamping, 0.03), 2: (qml.AmplitudeDamping, 0.05),
                    3: (qml.DepolarizingChannel, 0.02)}

    def _circuit(self, inputs, weights):
        for j in range(self.hidden_features):
            qml.RZ(inputs[..., j], wires=j)
            self._add_noise_ops(j, self._noise_table)
        qml.StronglyEntanglingLayers(weights, wires=range(self.hidden_features), imprimitive=qml.ops.CZ)
        return [qml.expval(qml.PauliZ(i)) for i in range(self.hidden_features)]

    def _circuit_descriptor(self):
        return _c.Circuit(n_qubits=self.hidden_features, encoding="rz", imprimitive="CZ", measure="expz",
                          n_rounds=1, n_blocks=1, sel_layers=self.qdepth)

    def _train_family(self):
        if type(self)._circuit is not QNN_noise._circuit:
            return None
        return self._circuit_descriptor(), self.linear_down, self.weights, self.linear_up

    def forward(self, x):
        b, c, w, h = x.shape
        flat = x.reshape(b, -1).to(self.linear_down.weight.device).to(torch.double)
        if self._fused_rounds_ok() and self.hidden_features <= 10:
            # inference: linear_down + circuit + linear_up in one launch
            circ = self._circuit_descriptor()
            out = _c.dense_forward(circ, flat, self.linear_down.weight, self.linear_down.bias,
                                   self.weights.reshape(circ.angles_shape), self.linear_up.weight,
                                   self.linear_up.bias)
            return out.view(b, c, w, h)
        reduced = self.linear_down(flat)
        ev = self.qnode(reduced, self.weights)
        if self.detach_quantum:
            ev = ev.detach()          # torch.tensor(qnode(...)) at reference :279 (finding F1)
        ev = ev.to(torch.double)
        return self.linear_up(ev).view(b, c, w, h)

    def fused_sample_steps(self, x, n_steps, goal, noise_factor=1.0):
        """n_steps bodies of Diffusion.sample in one launch; None when not applicable."""
        if not (self._fused_rounds_ok() and 2 <= self.hidden_features <= 10):
            return None
        b, c, w, h = x.shape
        circ = self._circuit_descriptor()
        flat = x.reshape(b, -1).to(self.linear_down.weight.device).to(torch.double)
        if flat.shape[1] > 2048 or flat.shape[1] != self.linear_up.weight.shape[0]:
            return None
        try:
            out = self._fused_sampler_launch(circ, flat, self.weights, n_steps, goal, noise_factor)
        except _capi.QiddmError as e:
            if e.code == -2:      # outside the fused sampler's range (e.g. tables beyond LDS): step by step
                return None
            raise
        return out.view(n_steps, b, c, w, h)

    def __repr__(self):
        return f"QNN(qdepth={self.qdepth}, features={self.hidden_features}, add_noise={self.add_noise})"

    def save_name(self) -> str:
        return f"QNN_linear_features={self.hidden_features}_qdepth={self.qdepth}_add_noise={self.add_noise}"


class QNN(QNN_noise):
    """Reference nn/qdense.py:310-386 (``QNN_noise`` without the noise switch)."""

    def __init__(self, input_dim, hidden_features, qdepth: int, detach_quantum=True) -> None:
        super().__init__(input_dim, hidden_features, qdepth, add_noise=0, detach_quantum=detach_quantum)

    def __repr__(self):
        return f"QNN(qdepth={self.qdepth}, features={self.hidden_features})"

    def save_name(self) -> str:
        return f"QNN_linear_features={self.hidden_features}_qdepth={self.qdepth}"


# ===========================================================================
# A3: PCA -> N x [L x (RZ + SEL(CZ, 2 layers))] -> probs, batched
# ===========================================================================
class differN_noise(_QuantumNet):
    """Reference nn/qdense.py:389-478.  ``(shape, spectrum_layer, N, add_noise=0)``."""

    _fusable_noise = (0, 1)   # add_noise=1 is PhaseShift right before probs: no effect (K9)

    def __init__(self, shape, spectrum_layer, N, add_noise=0) -> None:
        super().__init__()
        self._init_differn(shape, spectrum_layer, N, add_noise)
        self._make_qnode("default.qubit.torch", "backprop")

    def _init_differn(self, shape, spectrum_layer, N, add_noise):
        from sklearn.decomposition import PCA
        self.spectrum_layer = spectrum_layer
        self.N = N
        self.add_noise = add_noise
        self.width, self.height = _pair(shape)
        self.pixels = self.width * self.height
        self.wires = math.ceil(math.log2(self.pixels))
        self.pca = PCA(n_components=self.wires)
        weight_shape = (N, self.spectrum_layer, 2, self.wires, 3)
        self.weights = nn.Parameter(torch.randn(weight_shape, requires_grad=True) * 0.4)

    def _circuit(self, inputs, weights):
        for i in range(self.spectrum_layer):
            for j in range(self.wires):
                qml.RZ(inputs[:, j], wires=j)
            qml.StronglyEntanglingLayers(weights[i], wires=range(self.wires), imprimitive=qml.ops.CZ)
        for wire in range(self.wires):
            self._add_noise_ops(wire, {1: (qml.PhaseShift, 0.05), 2: (qml.AmplitudeDamping, 0.1),
                                       3: (qml.DepolarizingChannel, 0.02)})
        return qml.probs(wires=range(self.wires))

    def _post_process(self, probs):
        return torch.clamp(probs[:, : self.pixels] * self.pixels, 0, 1)

    def reduce(self, x):
        """Host-side PCA front-end, re-fit on every call as written (:456, finding F4)."""
        flat = x.reshape(x.shape[0], self.pixels)
        red = _pca.fit_transform(self.pca, flat)
        return red.to(torch.float32).to(next(self.parameters()).device)

    def forward_from_reduced(self, red):
        """Everything after the PCA (:464-472): the part that is parity-tested."""
        if self._fused_rounds_ok():
            circ = _c.Circuit(n_qubits=self.wires, encoding="rz", imprimitive="CZ", measure="probs",
                              n_rounds=self.N, n_blocks=self.spectrum_layer, sel_layers=2)
            if self.wires <= 10 and type(self)._post_process is differN_noise._post_process:
                # inference: all rounds AND the post-processing in one launch (the (B, 2^n) probabilities are never written)
                ang, table = self._gate_table(circ, self.weights)
                out = _c.run_forward_post(circ, red, ang, self.pixels, float(self.pixels), table=table)
                return out.reshape(red.shape[0], 1, self.width, self.height)
            p = _c.execute(circ, red, self.weights).to(torch.float64)
        else:
            p = red
            for n in range(self.N):
                p = self.qnode(p, self.weights[n])
        out = self._post_process(p)
        return out.reshape(red.shape[0], 1, self.width, self.height)

    def forward(self, x):
        return self.forward_from_reduced(self.reduce(x))

    def __repr__(self):
        return f"differN_old_pca={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"

    def save_name(self) -> str:
        return f"differN_old_pca={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}_noise{self.add_noise}"


class differN_noise_befor(differN_noise):
    """Reference nn/qdense.py:481-562 (the class the shipped Ray-Tune checkpoints were
    trained with; noise channels sit after every RZ)."""

    def __init__(self, shape, spectrum_layer, N, add_noise=0, device_type="default.qubit.torch") -> None:
        nn.Module.__init__(self)
        self._init_differn(shape, spectrum_layer, N, add_noise)
        self._make_qnode(device_type, "backprop")

    def _circuit(self, inputs, weights):
        for i in range(self.spectrum_layer):
            for j in range(self.wires):
                qml.RZ(inputs[:, j], wires=j)
                self._add_noise_ops(j, {1: (qml.PhaseDamping, 0.03), 2: (qml.AmplitudeDamping, 0.05),
                                        3: (qml.DepolarizingChannel, 0.02)})
            qml.StronglyEntanglingLayers(weights[i], wires=range(self.wires), imprimitive=qml.ops.CZ)
        return qml.probs(wires=range(self.wires))

    _fusable_noise = (0,)

    def __repr__(self):
        return f"differN_noise={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"

    def save_name(self) -> str:
        return f"differN_noise={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"


class differN_old_pca(differN_noise):
    """Reference nn/qdense.py:671-743 (``differN_noise`` without the noise switch)."""

    def __init__(self, shape, spectrum_layer, N) -> None:
        super().__init__(shape, spectrum_layer, N, add_noise=0)

    def save_name(self) -> str:
        return f"differN_old_pca={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"


# ===========================================================================
# A2: [PCA | linear_down] -> N x [L x (RZ + SEL(CZ, 2 layers))] -> <Z> -> linear_up
# ===========================================================================
class _QIDDMBase(_QuantumNet):
    _use_pca = False
    _noise_table = {1: (qml.PhaseDamping, 0.03), 2: (qml.AmplitudeDamping, 0.05),
                    3: (qml.DepolarizingChannel, 0.9)}

    def _init_qiddm(self, input_dim, hidden_features, spectrum_layer, N, add_noise, device_type,
                    detach_quantum):
        self.hidden_features = hidden_features
        self.spectrum_layer = spectrum_layer
        self.N = N
        self.add_noise = add_noise
        self.detach_quantum = detach_quantum
        if self._use_pca:
            from sklearn.decomposition import PCA
            self.pca = PCA(n_components=hidden_features)
        else:
            self.linear_down = nn.Linear(input_dim, hidden_features)
        self.linear_up = nn.Linear(hidden_features, input_dim)
        weight_shape1 = (N, self.spectrum_layer, 2, hidden_features, 3)
        self.weights1 = nn.Parameter(torch.randn(weight_shape1, requires_grad=True) * 0.4)
        self._make_qnode(device_type, "parameter-shift")

    def _circuit(self, inputs, weights1):
        for i in range(self.spectrum_layer):
            for j in range(self.hidden_features):
                qml.RZ(inputs[..., j], wires=j)
                self._add_noise_ops(j, self._noise_table)
            qml.StronglyEntanglingLayers(weights1[i], wires=range(self.hidden_features),
                                         imprimitive=qml.ops.CZ)
        return [qml.expval(qml.PauliZ(i)) for i in range(self.hidden_features)]

    def reduce(self, flat):
        if self._use_pca:
            red = _pca.fit_transform(self.pca, flat)              # re-fit per call (F4)
            return red.to(self.linear_up.weight.device).to(self.linear_up.weight.dtype)
        return self.linear_down(flat)

    def quantum_rounds(self, red):
        """The N chained QNode rounds (reference :1437-1441 / :1631-1635), one wavefront per
        sample instead of the per-sample Python loop."""
        if self._fused_rounds_ok():
            circ = _c.Circuit(n_qubits=self.hidden_features, encoding="rz", imprimitive="CZ",
                              measure="expz", n_rounds=self.N, n_blocks=self.spectrum_layer, sel_layers=2)
            return _c.execute(circ, red, self.weights1).to(torch.float64)
        x = red
        for n in range(self.N):
            x = self.qnode(x, self.weights1[n])
            x = (x.detach() if self.detach_quantum else x).to(torch.float64)   # finding F1
        return x

    def _train_family(self):
        if self._use_pca or not hasattr(self, "linear_down") or type(self)._circuit is not _QIDDMBase._circuit \
                or type(self).forward is not _QIDDMBase.forward:
            return None
        circ = _c.Circuit(n_qubits=self.hidden_features, encoding="rz", imprimitive="CZ", measure="expz",
                          n_rounds=self.N, n_blocks=self.spectrum_layer, sel_layers=2)
        return circ, self.linear_down, self.weights1, self.linear_up

    def fused_sample_steps(self, x, n_steps, goal, noise_factor=1.0):
        """n_steps bodies of Diffusion.sample in one launch; None when not applicable."""
        if not (self._fused_rounds_ok() and not self._use_pca and hasattr(self, "linear_down")
                and 2 <= self.hidden_features <= 10 and type(self)._circuit is _QIDDMBase._circuit):
            return None
        b, c, w, h = x.shape
        flat = x.reshape(b, -1)
        if flat.shape[1] > 2048 or flat.shape[1] != self.linear_up.weight.shape[0]:
            return None
        circ = _c.Circuit(n_qubits=self.hidden_features, encoding="rz", imprimitive="CZ", measure="expz",
                          n_rounds=self.N, n_blocks=self.spectrum_layer, sel_layers=2)
        try:
            out = self._fused_sampler_launch(circ, flat, self.weights1, n_steps, goal, noise_factor)
        except _capi.QiddmError as e:
            if e.code == -2:      # outside the fused sampler's range: step by step
                return None
            raise
        return out.to(self.linear_up.weight.dtype).view(n_steps, b, c, w, h)

    def forward(self, x):
        b, c, w, h = x.shape
        if self._fused_rounds_ok() and not self._use_pca and self.hidden_features <= 10:
            # inference: linear_down + N chained rounds + linear_up in one launch
            circ = _c.Circuit(n_qubits=self.hidden_features, encoding="rz", imprimitive="CZ",
                              measure="expz", n_rounds=self.N, n_blocks=self.spectrum_layer, sel_layers=2)
            out = _c.dense_forward(circ, x.reshape(b, -1), self.linear_down.weight, self.linear_down.bias,
                                   self.weights1, self.linear_up.weight, self.linear_up.bias)
            return out.to(self.linear_up.weight.dtype).view(b, c, w, h)
        red = self.reduce(x.reshape(b, -1))
        ev = self.quantum_rounds(red)
        ev = ev.to(self.linear_up.weight.device).to(self.linear_up.weight.dtype)
        return self.linear_up(ev).view(b, c, w, h)


class QIDDM_LL_noise(_QIDDMBase):
    """Reference nn/qdense.py:1567-1660."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, add_noise=0,
                 device_type="lightning.qubit", detach_quantum=True) -> None:
        super().__init__()
        self._init_qiddm(input_dim, hidden_features, spectrum_layer, N, add_noise, device_type, detach_quantum)

    def __repr__(self):
        return (f"QIDDM_LL_noise(qlayer={self.spectrum_layer}, features={self.hidden_features}, "
                f"N={self.N}, add_noise={self.add_noise})")

    def save_name(self) -> str:
        return f"QIDDM_LL_noise={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_LL_relu_noise(QIDDM_LL_noise):
    """Reference nn/qdense.py:1469-1564: constructs an ``nn.ReLU`` (:1500) that is never
    applied -- identical mathematics and identical ``save_name``."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.relu = nn.ReLU()


QIDDM_L = QIDDM_LL_noise   # imported by the reference drivers but never defined there


class QIDDM_PL_noise(_QIDDMBase):
    """Reference nn/qdense.py:1371-1466 (PCA front-end)."""

    _use_pca = True

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, add_noise=0,
                 device_type="lightning.qubit", detach_quantum=True) -> None:
        super().__init__()
        self._init_qiddm(input_dim, hidden_features, spectrum_layer, N, add_noise, device_type, detach_quantum)

    def __repr__(self):
        return (f"QIDDM_PL_noise(qlayer={self.spectrum_layer}, features={self.hidden_features}, "
                f"N={self.N}, add_noise={self.add_noise})")

    def save_name(self) -> str:
        return f"QIDDM_PL_noise={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_PL(_QIDDMBase):
    """Reference nn/qdense.py:1271-1368."""

    _use_pca = True

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, detach_quantum=True) -> None:
        super().__init__()
        self._init_qiddm(input_dim, hidden_features, spectrum_layer, N, 0, "lightning.qubit", detach_quantum)

    def __repr__(self):
        return f"QIDDM_PL(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_PL={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"
