"""The remaining 14 classes of reference nn/qdense.py: recombinations of the same circuit templates
({RZ | RY | pi/2*RZ} re-uploading x SEL(CZ) x {probs | <Z>}) with other classical front-/back-ends
(PCA, strided Conv2d + mean, BatchNorm1d, PCA inverse transform).  Same constructor orders,
parameter names, ``save_name()`` strings; per-sample Python loops are batched (one wavefront per
sample); ``torch.tensor(...)`` / ``.clone().detach()`` wrappers of the reference are kept as
``detach_quantum`` (finding F1)."""
from __future__ import annotations

import math
import pickle

import torch
import torch.nn as nn

from .. import pca as _pca
from .. import qml
from .qdense import _QIDDMBase, _QuantumNet, _pair, differN_noise


def _sel_cz_expz(self, inputs, weights, enc=qml.RZ, scale=1.0):
    n = self.hidden_features
    xin = inputs * scale if scale != 1.0 else inputs      # one tensor: every block re-uploads the same views
    for i in range(self.spectrum_layer):
        for j in range(n):
            enc(xin[..., j], wires=j)
            self._add_noise_ops(j, getattr(self, "_noise_table", {}))
        qml.StronglyEntanglingLayers(weights[i], wires=range(n), imprimitive=qml.ops.CZ)


class QIDDM_PL_noise1(_QIDDMBase):
    """Reference nn/qdense.py:565-668: as QIDDM_PL_noise but the data is re-uploaded with RY (:600-602)."""

    _use_pca = True

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, add_noise=0,
                 device_type="lightning.qubit", detach_quantum=True) -> None:
        super().__init__()
        self._init_qiddm(input_dim, hidden_features, spectrum_layer, N, add_noise, device_type, detach_quantum)

    def _circuit(self, inputs, weights1):
        _sel_cz_expz(self, inputs, weights1, enc=qml.RY)
        return [qml.expval(qml.PauliZ(i)) for i in range(self.hidden_features)]

    def _fused_rounds_ok(self) -> bool:
        return False        # the fused multi-round launch is specialised on the RZ encoding

    def __repr__(self):
        return (f"QIDDM_PL_noise(qlayer={self.spectrum_layer}, features={self.hidden_features}, "
                f"N={self.N}, add_noise={self.add_noise})")

    def save_name(self) -> str:
        return f"QIDDM_PL_noise={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_PL_old(_QIDDMBase):
    """Reference nn/qdense.py:1176-1268."""

    _use_pca = True

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, detach_quantum=True) -> None:
        super().__init__()
        self._init_qiddm(input_dim, hidden_features, spectrum_layer, N, 0, "lightning.qubit", detach_quantum)

    def __repr__(self):
        return f"QIDDM(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_PL_old_q={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_LL_old(_QIDDMBase):
    """Reference nn/qdense.py:1873-1968."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, detach_quantum=True) -> None:
        super().__init__()
        self._init_qiddm(input_dim, hidden_features, spectrum_layer, N, 0, "lightning.qubit", detach_quantum)

    def __repr__(self):
        return f"QIDDM(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_linear_features={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_bias_false(_QuantumNet):
    """Reference nn/qdense.py:1971-2074: bias-free linears, three SEL layers per block
    (weights1: (N, L, 3, n, 3))."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, detach_quantum=True) -> None:
        super().__init__()
        self.hidden_features, self.spectrum_layer, self.N = hidden_features, spectrum_layer, N
        self.detach_quantum = detach_quantum
        self.linear_down = nn.Linear(input_dim, hidden_features, bias=False)
        self.linear_up = nn.Linear(hidden_features, input_dim, bias=False)
        self.weights1 = nn.Parameter(torch.randn((N, spectrum_layer, 3, hidden_features, 3), requires_grad=True) * 0.4)
        self._make_qnode("lightning.qubit", "parameter-shift")

    def _circuit(self, inputs, weights1):
        _sel_cz_expz(self, inputs, weights1)
        return [qml.expval(qml.PauliZ(i)) for i in range(self.hidden_features)]

    def forward(self, x):
        b, c, w, h = x.shape
        flat = x.reshape(b, -1).to(self.linear_down.weight.device).to(self.linear_down.weight.dtype)
        red = self.linear_down(flat)
        for n in range(self.N):
            red = self.qnode(red, self.weights1[n])
            red = red.detach() if self.detach_quantum else red
        red = red.to(self.linear_up.weight.dtype)
        return self.linear_up(red).view(b, c, w, h)

    def __repr__(self):
        return f"QIDDM(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_linear_features={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_L_B(QIDDM_bias_false):
    """Reference nn/qdense.py:2077-2179: BatchNorm1d in front of every round, ``default.qubit.jax`` +
    backprop, QNode outputs NOT detached (gradients reach the quantum weights here)."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int) -> None:
        nn.Module.__init__(self)
        self.hidden_features, self.spectrum_layer, self.N = hidden_features, spectrum_layer, N
        self.linear_down = nn.Linear(input_dim, hidden_features)
        self.batchnorm = nn.BatchNorm1d(hidden_features)
        self.linear_up = nn.Linear(hidden_features, input_dim)
        self.weights1 = nn.Parameter(torch.randn((N, spectrum_layer, 3, hidden_features, 3), requires_grad=True) * 0.4)
        self._make_qnode("default.qubit.jax", "backprop")

    def forward(self, x):
        b, c, w, h = x.shape
        red = self.linear_down(x.reshape(b, -1).to(self.linear_down.weight.dtype))
        for n in range(self.N):
            red = self.batchnorm(red.to(self.batchnorm.weight.dtype)).to(self.weights1.dtype)
            red = self.qnode(red, self.weights1[n])
        return self.linear_up(red.reshape(b, -1).to(self.linear_up.weight.dtype)).view(b, c, w, h)

    def __repr__(self):
        return f"QIDDM_L_B(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_linear_batch_features={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class _ConvFront:
    """``Conv2d(1 -> n, k=3, stride=2, pad=1)`` then the spatial mean: the (b, n) angles of the
    conv-front-end classes (reference nn/qdense.py:853, 899-901)."""

    def _conv_reduce(self, x):
        y = self.conv_layer(x)
        return y.reshape(x.shape[0], y.shape[1], -1).mean(dim=2)


class QIDDM_CL_new(_QIDDMBase, _ConvFront):
    """Reference nn/qdense.py:1014-1101."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, detach_quantum=True) -> None:
        nn.Module.__init__(self)
        self.hidden_features, self.spectrum_layer, self.N = hidden_features, spectrum_layer, N
        self.add_noise, self.detach_quantum = 0, detach_quantum
        self.conv_layer = nn.Conv2d(in_channels=1, out_channels=hidden_features, kernel_size=3, stride=2, padding=1)
        self.linear_up = nn.Linear(hidden_features, input_dim)
        self.weights1 = nn.Parameter(torch.randn((N, spectrum_layer, 2, hidden_features, 3), requires_grad=True) * 0.4)
        self._make_qnode("lightning.qubit", "parameter-shift")

    def forward(self, x):
        b, c, w, h = x.shape
        ev = self.quantum_rounds(self._conv_reduce(x))
        ev = ev.to(self.linear_up.weight.device).to(self.linear_up.weight.dtype)
        return self.linear_up(ev).view(b, c, w, h)

    def __repr__(self):
        return f"QIDDM(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_CL_new_q={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_CL_old(QIDDM_CL_new):
    """Reference nn/qdense.py:1104-1173.  As written it hands the whole (b, n) batch to a per-sample
    circuit (``inputs.flatten()[j]``), which only works for b == 1; implemented with the per-sample
    semantics of QIDDM_CL_new and, as in the reference, without the ``torch.tensor`` detach."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int) -> None:
        super().__init__(input_dim, hidden_features, spectrum_layer, N, detach_quantum=False)

    def save_name(self) -> str:
        return f"QIDDM_CL_old_q={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_PP_noise(_QIDDMBase):
    """Reference nn/qdense.py:1663-1753: PCA in, ``pca.inverse_transform`` out; the only parameters are
    the circuit weights."""

    _use_pca = True

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, add_noise=0,
                 device_type="lightning.qubit", detach_quantum=True) -> None:
        nn.Module.__init__(self)
        from sklearn.decomposition import PCA
        self.hidden_features, self.spectrum_layer, self.N = hidden_features, spectrum_layer, N
        self.add_noise, self.detach_quantum = add_noise, detach_quantum
        self.pca = PCA(n_components=hidden_features)
        self.weights1 = nn.Parameter(torch.randn((N, spectrum_layer, 2, hidden_features, 3), requires_grad=True) * 0.4)
        self._make_qnode(device_type, "parameter-shift")

    def forward(self, x):
        b, c, w, h = x.shape
        red = _pca.fit_transform(self.pca, x.reshape(b, -1))
        red = red.to(self.weights1.device).to(self.weights1.dtype)
        ev = self.quantum_rounds(red)
        restored = _pca.inverse_transform(self.pca, ev.reshape(b, -1))
        return restored.detach().to(device=x.device, dtype=x.dtype).requires_grad_(True).view(b, c, w, h)

    def __repr__(self):
        return (f"QIDDM_PP_noise(qlayer={self.spectrum_layer}, features={self.hidden_features}, "
                f"N={self.N}, add_noise={self.add_noise})")

    def save_name(self) -> str:
        return f"QIDDM_PP_noise={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_PP_old(_QIDDMBase):
    """Reference nn/qdense.py:1756-1870: PCA(2n) fitted once, BatchNorm1d, linear_down (2n -> n), rounds,
    linear_up (n -> 2n), PCA inverse; ``save_model`` pickles the PCA next to the state dict."""

    def __init__(self, input_dim, hidden_features, spectrum_layer, N: int, detach_quantum=True) -> None:
        nn.Module.__init__(self)
        self.hidden_features, self.spectrum_layer, self.N = hidden_features, spectrum_layer, N
        self.input_dim, self.add_noise, self.detach_quantum = input_dim, 0, detach_quantum
        self.pca = None
        self.batch_norm = nn.BatchNorm1d(2 * hidden_features)
        self.linear_down = nn.Linear(2 * hidden_features, hidden_features)
        self.linear_up = nn.Linear(hidden_features, 2 * hidden_features)
        self.weights1 = nn.Parameter(torch.randn((N, spectrum_layer, 2, hidden_features, 3), requires_grad=True) * 0.4)
        self._make_qnode("lightning.qubit", "parameter-shift")

    def forward(self, x):
        from sklearn.decomposition import PCA
        b, c, w, h = x.shape
        flat = x.reshape(b, -1)
        if self.pca is None:
            self.pca = PCA(n_components=2 * self.hidden_features)
            _pca.fit(self.pca, flat)
        red = _pca.transform(self.pca, flat).detach().to(device=x.device, dtype=x.dtype).requires_grad_(True)
        red = self.linear_down(self.batch_norm(red))
        ev = self.quantum_rounds(red).to(x.dtype)
        up = self.linear_up(ev).reshape(b, -1)
        restored = _pca.inverse_transform(self.pca, up)
        return restored.detach().to(device=x.device, dtype=x.dtype).requires_grad_(True).view(b, c, w, h)

    def __repr__(self):
        return f"QIDDM_PP(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"QIDDM_PP_features={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"

    def save_model(self, path):
        model_dict = {"model_state_dict": self.state_dict()}
        if self.pca is not None:
            model_dict["pca_state"] = pickle.dumps(self.pca)
        torch.save(model_dict, path)

    def load_model(self, path):
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
        self.load_state_dict(checkpoint["model_state_dict"])
        if "pca_state" in checkpoint:
            self.pca = pickle.loads(checkpoint["pca_state"])


# ---------------------------------------------------------------------------
# probs-chained nets on ceil(log2(pixels)) wires
# ---------------------------------------------------------------------------
class differN_new_pca(differN_noise):
    """Reference nn/qdense.py:747-835: per-sample processing and ``_post_process`` (slice, x pixels,
    clamp) BETWEEN the rounds -- the next round's angles are the post-processed pixels."""

    def __init__(self, shape, spectrum_layer, N) -> None:
        super().__init__(shape, spectrum_layer, N, add_noise=0)

    def _fused_rounds_ok(self) -> bool:
        return False

    def forward_from_reduced(self, red):
        x = red
        for n in range(self.N):
            x = self._post_process(self.qnode(x, self.weights[n]))
        return x.reshape(red.shape[0], 1, self.width, self.height)

    def __repr__(self):
        return f"differN_new_pca={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"

    save_name = __repr__


class differN_old_conv(differN_noise, _ConvFront):
    """Reference nn/qdense.py:939-1011: strided-conv front-end, raw-probability chaining, batched."""

    def __init__(self, shape, spectrum_layer, N) -> None:
        nn.Module.__init__(self)
        from sklearn.decomposition import PCA
        self.spectrum_layer, self.N, self.add_noise = spectrum_layer, N, 0
        self.width, self.height = _pair(shape)
        self.pixels = self.width * self.height
        self.wires = math.ceil(math.log2(self.pixels))
        self.pca = PCA(n_components=self.wires)          # constructed but unused, as in the reference
        # RNG order of the reference: conv_layer first, then the circuit weights (nn/qdense.py:953-962)
        self.conv_layer = nn.Conv2d(in_channels=1, out_channels=self.wires, kernel_size=3, stride=2, padding=1)
        self.weights = nn.Parameter(torch.randn((N, spectrum_layer, 2, self.wires, 3), requires_grad=True) * 0.4)
        self._make_qnode("default.qubit.torch", "backprop")

    def reduce(self, x):
        return self._conv_reduce(x)

    def __repr__(self):
        return f"differN_old_conv={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"

    save_name = __repr__


class differN_new_conv(differN_old_conv):
    """Reference nn/qdense.py:838-936: conv front-end + the per-round post-processing of differN_new_pca."""

    _fused_rounds_ok = differN_new_pca._fused_rounds_ok
    forward_from_reduced = differN_new_pca.forward_from_reduced

    def __repr__(self):
        return f"differN_new_conv={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"

    save_name = __repr__


class QIDDM_A_sameN(differN_noise):
    """Reference nn/qdense.py:2276-2342: no front-end (the first `wires` pixels are the angles), ONE
    weight tensor (L, 2, n, 3) shared by all N rounds."""

    def __init__(self, shape, spectrum_layer, N) -> None:
        nn.Module.__init__(self)
        self.spectrum_layer, self.N, self.add_noise = spectrum_layer, N, 0
        self.width, self.height = _pair(shape)
        self.pixels = self.width * self.height
        self.wires = math.ceil(math.log2(self.pixels))
        self.weights = nn.Parameter(torch.randn((spectrum_layer, 2, self.wires, 3), requires_grad=True) * 0.4)
        self._make_qnode("default.qubit.torch", "backprop")

    def reduce(self, x):
        return x.reshape(x.shape[0], self.pixels)

    def _fused_rounds_ok(self) -> bool:
        return False

    def forward_from_reduced(self, red):
        p = red
        for _ in range(self.N):
            p = self.qnode(p, self.weights)
        return self._post_process(p).reshape(red.shape[0], 1, self.width, self.height)

    def __repr__(self):
        return f"QIDDM_A_sameN={self.spectrum_layer}_N={self.N}_w{self.width}_h{self.height}"

    save_name = __repr__


class QIDDM_A_differN_basePL(_QuantumNet):
    """Reference nn/qdense.py:2182-2273: PCA -> N x [L x (RZ(pi/2 x) + SEL(CZ, 2 layers))] -> probs,
    post-processed (and detached, :2244-2246) between the rounds; output = the last post-processed
    pixels.  ``(input_dim, spectrum_layer, N)`` with input_dim the image side."""

    _name = "QIDDM_pca_features"

    def __init__(self, input_dim, spectrum_layer, N: int, detach_quantum=True) -> None:
        super().__init__()
        from sklearn.decomposition import PCA
        self.spectrum_layer, self.N, self.detach_quantum = spectrum_layer, N, detach_quantum
        self.width = self.height = input_dim
        self.pixels = self.width * self.height
        self.hidden_features = math.ceil(math.log2(self.pixels))
        self.pca = PCA(n_components=self.hidden_features)
        self.weights1 = nn.Parameter(torch.randn((N, spectrum_layer, 2, self.hidden_features, 3), requires_grad=True) * 0.4)
        self._make_qnode("lightning.qubit", "parameter-shift")

    def _circuit(self, inputs, weights1):
        _sel_cz_expz(self, inputs, weights1, scale=math.pi * 0.5)
        return qml.probs(wires=range(self.hidden_features))

    def _post_process(self, probs):
        return torch.clamp(probs[..., : self.pixels] * self.pixels, 0, 1)

    def forward_from_reduced(self, red):
        x = red
        for n in range(self.N):
            p = self.qnode(x, self.weights1[n])
            p = p.detach() if self.detach_quantum else p
            x = self._post_process(p)
        return x

    def forward(self, x):
        b, c, w, h = x.shape
        red = _pca.fit_transform(self.pca, x.reshape(b, -1))
        red = red.to(x.device).to(x.dtype)
        return self.forward_from_reduced(red).reshape(b, c, w, h)

    def __repr__(self):
        return f"QIDDM(qlayer={self.spectrum_layer}, features={self.hidden_features}, N={self.N})"

    def save_name(self) -> str:
        return f"{self._name}={self.hidden_features}_L={self.spectrum_layer}_N={self.N}"


class QIDDM_A_differN_NEW(QIDDM_A_differN_basePL):
    """Reference nn/qdense.py:2345-2436 (same network, other ``save_name``)."""

    _name = "QIDDM_pca_new"
