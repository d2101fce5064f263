"""Quantum convolution (reference nn/qconv.py:8-126, exported as ``QConv2d`` at :307).

Implements the *intended* layer (SURVEY.md finding F3): the reference's
``forward`` (:71-87) never calls ``self.qnode``, so as checked in it returns
``ceil(C k^2 / 2)`` post-processed *input patches* instead of ``out_channels``
circuit outputs and crashes the BatchNorm that follows it in ``unet_simple``.  The
only meaningful reading -- ``x = self.qnode(x)`` between :78 and :79 -- is what
runs here:

    unfold -> (+0.1) -> AmplitudeEmbedding(pad 0.5, normalize) ->
    StronglyEntanglingLayers(pi*tanh(W), CNOT) -> probs -> *D/2, clamp, [::2], [:C_out]

one circuit per output pixel, one wavefront per circuit.
"""
from __future__ import annotations

import math
import warnings

import torch

from .. import circuit as _c
from .. import qml
from .qdense import _qw_tanh
from .utils import unfold_patches


class QConv2d(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), padding=1, qdepth=2):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size if isinstance(kernel_size, tuple) else (kernel_size, kernel_size)
        self.padding = padding if isinstance(padding, tuple) else (padding, padding)
        self.unfold = torch.nn.Unfold(kernel_size=kernel_size, padding=padding).double()
        wires_for_inp = math.ceil(math.log2(self.kernel_size[0] * self.kernel_size[1] * in_channels))
        wires_for_out = math.ceil(math.log2(out_channels))
        self.wires = max(wires_for_inp, wires_for_out, 1)
        if self.wires > 10:
            warnings.warn(f"Too many wires ({self.wires}). This might cause performance issues.")
        template_shape = qml.StronglyEntanglingLayers.shape(n_layers=qdepth, n_wires=self.wires)
        w = torch.rand(template_shape, dtype=torch.double, requires_grad=True)
        self.weights = torch.nn.Parameter(w * math.pi - math.pi / 2)
        self.qdev = qml.device("default.qubit.torch", wires=self.wires)
        self.qnode = qml.QNode(func=self._circuit, device=self.qdev, cache=True, cachesize=int(1e6),
                               interface="torch", diff_method="backprop")
        self.sample_qnode = None
        self.sample_matrix = None
        self._own_qnode = self.qnode

    def _circuit(self, features):
        qml.AmplitudeEmbedding(features=features.double(), wires=range(self.wires), pad_with=0.5,
                               normalize=True)
        qml.StronglyEntanglingLayers(_qw_tanh(self.weights.double()), wires=range(self.wires))
        return qml.probs(wires=range(self.wires))

    def _post_process(self, quantum_probs):
        p = torch.clamp(quantum_probs * quantum_probs.shape[-1] * 0.5, 0.0, 1.0)
        return p[:, ::2][:, : self.out_channels].double()

    def forward(self, x):
        b, c, h_in, w_in = x.shape
        assert c == self.in_channels, f"Expected {self.in_channels} channels, got {c}"
        h_out = h_in + 2 * self.padding[0] - self.kernel_size[0] + 1
        w_out = w_in + 2 * self.padding[1] - self.kernel_size[1] + 1
        if (not torch.is_grad_enabled() and self.qnode is self._own_qnode and self.wires <= 12 and not self.training
                and x.is_cuda and _c._default_precision == "f32" and 2 * self.out_channels <= 2 ** self.wires):
            # eval mode (reference :92-126), up to C4's 12 wires: the cached circuit unitary, then one GEMM
            return _c.qconv_unitary_forward(x, self._eval_unitary(), self.wires, self.out_channels,
                                            self.kernel_size, self.padding, **self._packed_args())
        if (not torch.is_grad_enabled() and self.qnode is self._own_qnode and self.wires <= 10
                and 2 * self.out_channels <= 2 ** self.wires):
            if not self.training and x.is_cuda and _c._default_precision == "f32":
                # eval mode (reference :92-126): the cached circuit unitary, then one GEMM on the matrix cores
                return _c.qconv_unitary_forward(x, self._eval_unitary(), self.wires, self.out_channels,
                                                self.kernel_size, self.padding, **self._packed_args())
            # inference: unfold + embedding + circuit + post-processing in one launch
            return _c.qconv_forward(x, _qw_tanh(self.weights.detach().double()), self.wires,
                                    self.out_channels, self.kernel_size, self.padding)
        if (self.qnode is self._own_qnode and x.is_cuda and _c._default_precision == "f32"
                and _c.qconv_unitary_trainable(self.wires, self.in_channels, self.kernel_size, self.out_channels)):
            # training through the circuit unitary: GEMM forward, thin-product backward, one adjoint sweep per
            # output channel (the circuit does not depend on the data)
            # (the angle map runs on the weight-gradient stream: its backward, and the adjoint sweeps in front of it,
            #  then overlap the rest of the network's backward -- circuit.on_weight_grad_stream)
            angles = _c.on_weight_grad_stream(lambda w: _qw_tanh(w.double()), self.weights)
            return _c.qconv_unitary_execute(x.double(), angles, self.wires, self.out_channels, self.kernel_size,
                                            self.padding)
        if (self.qnode is self._own_qnode and self.wires <= 10 and x.is_cuda
                and 2 * self.out_channels <= 2 ** self.wires):
            # training: the same fused launch, differentiable (adjoint sweep per output pixel + fold)
            return _c.qconv_execute(x.double(), _qw_tanh(self.weights.double()), self.wires, self.out_channels,
                                    self.kernel_size, self.padding)
        # self.unfold's columns, one row per output pixel (a strided view + one copy; nn/utils.unfold_patches)
        feats = unfold_patches(x.double(), self.kernel_size, self.padding) + 0.1
        y = self._post_process(self.qnode(feats))                        # ((b h w), C_out)
        return y.reshape(b, h_out, w_out, -1).permute(0, 3, 1, 2).contiguous()

    def train_forward_bn(self, x, bn):
        """``bn(self(x))`` as one autograd node when this layer trains through its circuit unitary and ``bn`` is a float64
        training-mode BatchNorm2d on the device (``circuit.qconv_bn_train``: the BatchNorm backward's transform pass is
        folded into the convolution's backward); None otherwise -- the caller then runs the two modules."""
        if not (torch.is_grad_enabled() and self.qnode is self._own_qnode and x.is_cuda and x.dim() == 4
                and x.numel() > 0 and x.shape[1] == self.in_channels and _c._default_precision == "f32"
                and _c._QCONV_BN_FUSED
                and _c.qconv_unitary_trainable(self.wires, self.in_channels, self.kernel_size, self.out_channels)
                and _c.batch_norm_eligible(bn, self.out_channels)
                and _c.qconv_bn_foldable(tuple(x.shape), self.wires, self.out_channels, self.kernel_size, self.padding)):
            return None
        angles = _c.on_weight_grad_stream(lambda w: _qw_tanh(w.double()), self.weights)
        return _c.qconv_bn_train(x.double(), angles, bn, self.wires, self.out_channels, self.kernel_size, self.padding)

    def eval_forward(self, x, upsample2x=False, batch_norm=None):
        """Inference through the cached unitary + GEMM with the layer's ``unet_simple`` neighbours folded in
        (``upsample2x``: the bilinear x2 in front of an ``up_conv``; ``batch_norm``: the eval-mode BatchNorm2d behind
        a ``net`` convolution).  None when this layer or the call is outside that route."""
        if torch.is_grad_enabled() or self.training or self.qnode is not self._own_qnode or not x.is_cuda \
                or self.wires > 12 or 2 * self.out_channels > 2 ** self.wires or x.shape[1] != self.in_channels:
            return None
        if _c._default_precision != "f32":
            return None      # the GEMM multiplies in float32; the float64 setting keeps the float64 circuit kernel
        if batch_norm is not None and (batch_norm.training or batch_norm.running_mean is None
                                       or batch_norm.num_features != self.out_channels):
            return None
        return _c.qconv_unitary_forward(x, self._eval_unitary(), self.wires, self.out_channels, self.kernel_size,
                                        self.padding, upsample2x=upsample2x, batch_norm=batch_norm, **self._packed_args())

    def _packed_args(self):
        """The layer's own packed-operand cache for ``circuit.qconv_unitary_forward`` (valid while ``_sample_stamp``, the
        stamp of the weights the cached unitary was built from, stands): call AFTER ``_eval_unitary()``."""
        if not hasattr(self, "_packed_operand"):
            self._packed_operand = {}
        return {"packed": self._packed_operand, "packed_key": getattr(self, "_sample_stamp", None)}

    def _eval_unitary(self):
        """``sample_matrix`` of the reference (:96-103): the (D, D) complex128 matrix of
        ``StronglyEntanglingLayers(pi*tanh(weights))``, rebuilt when the weights changed since it was taken."""
        stamp = (self.weights._version, self.weights.data_ptr(), self.weights.device)
        if self.sample_matrix is None or getattr(self, "_sample_stamp", None) != stamp:
            self.sample_matrix = _c.circuit_unitary(_qw_tanh(self.weights.detach().double()), self.wires, "CNOT")
            self._sample_stamp = stamp
        return self.sample_matrix

    def train(self, mode=True):
        """Leaving eval mode drops the cached unitary, as the reference does (:123-125); entering it lets the
        next forward build it (the reference builds it here, which needs the weights on the device already)."""
        super().train(mode)
        if mode:
            self.sample_qnode = None
            self.sample_matrix = None
        return self

    def __repr__(self):
        return (f"QConv2d({self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, "
                f"padding={self.padding}, wires={self.wires})")


_QConv2d_FAST = QConv2d
