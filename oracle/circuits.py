"""Oracle restatement of the reference's circuit templates and layer forwards.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  The RZ / SEL-CZ /
<Z> templates are **pinned** to reference-held outputs
(``tests/test_oracle_reference_runs.py``); the amplitude / SEL-CNOT / probs
templates are **parity unpinned** (no reference outputs exist for them); the
layer glue follows the reference lines cited per function.

``run_circuit`` is the generic executor; the five templates of SURVEY.md
section 8a are all of the form

    for round in N:                                   (chained QNode calls)
        [AmplitudeEmbedding | nothing]
        for block in L:
            [RZ(x_j) | RY(x_j) | nothing on every wire j]  (data re-upload)
            StronglyEntanglingLayers(W[round, block] : (S, n, 3), CNOT|CZ)
        probs | <Z_i>
        next round's x = first n entries of this round's output
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import statevector as sv


@dataclass
class Spec:
    n: int
    encoding: str = "rz"          # "amplitude" | "rz" | "ry" | "none"
    imprimitive: str = "CZ"       # "CNOT" | "CZ"
    measure: str = "expz"         # "probs" | "expz"
    enc_scale: float = 1.0        # angle scale for rz encoding
    enc_offset: float = 0.0       # added to features before amplitude embedding
    pad_with: float = 0.0
    ry_once: bool = True          # AngleEmbedding(Y) happens once, before block 0


def run_round(spec: Spec, inputs, weights):
    """One QNode evaluation.  ``weights``: (L, S, n, 3).  ``inputs``: (B, F).

    Follows the ``_circuit`` bodies at ``nn/qdense.py:40-47`` (amplitude/CNOT/
    probs), ``:164-183`` (RY/CNOT/probs), ``:249-265`` (RZ/CZ/<Z>),
    ``:422-441`` and ``:1403-1421`` (re-uploading RZ/CZ), ``nn/qconv.py:51-56``.
    """
    n = spec.n
    x = torch.as_tensor(inputs, dtype=sv.RDT)
    if x.dim() == 1:
        x = x.unsqueeze(0)
    b = x.shape[0]
    w = torch.as_tensor(weights).to(sv.RDT) if not torch.is_tensor(weights) else weights.to(sv.RDT)
    assert w.dim() == 4 and w.shape[2] == n and w.shape[3] == 3, w.shape
    if spec.encoding == "amplitude":
        st = sv.amplitude_embedding(x + spec.enc_offset, n, pad_with=spec.pad_with, normalize=True)
    else:
        st = sv.zero_state(b, n)
    for blk in range(w.shape[0]):
        if spec.encoding == "rz":
            st = sv.angle_embedding_rz(st, x, n, scale=spec.enc_scale)
        elif spec.encoding in ("ry", "ry_blocks") and (blk == 0 or spec.encoding == "ry_blocks" or not spec.ry_once):
            st = sv.angle_embedding_ry(st, x * spec.enc_scale, n)
        st = sv.strongly_entangling_layers(st, w[blk], n, spec.imprimitive)
    if spec.measure == "probs":
        return sv.probs(st)
    return sv.expval_z(st, n)


def run_circuit(spec: Spec, inputs, weights):
    """N chained rounds; ``weights``: (N, L, S, n, 3)  (reference loops
    ``nn/qdense.py:464-465`` and ``:1631-1635``)."""
    w = weights if torch.is_tensor(weights) else torch.as_tensor(weights)
    x = inputs
    out = None
    for r in range(w.shape[0]):
        out = run_round(spec, x, w[r])
        x = out
    return out


def gate_count(spec: Spec, n_rounds: int, n_blocks: int, sel_layers: int) -> int:
    """Gate applications per sample per forward, counted as SURVEY.md section 8a:
    Rot = 1, each CZ/CNOT = 1, each RZ/RY encoder = 1, embedding = 1."""
    n = spec.n
    per_block = sel_layers * (n + (n if n > 1 else 0))
    if spec.encoding in ("rz",):
        per_block += n
    g = n_rounds * n_blocks * per_block
    if spec.encoding == "ry":
        g += n_rounds * n
    if spec.encoding == "ry_blocks":
        g += n_rounds * n_blocks * n
    if spec.encoding == "amplitude":
        g += n_rounds
    return g


# ---------------------------------------------------------------------------
# layer forwards (float64), one function per reference class family
# ---------------------------------------------------------------------------
def qw_map_tanh(w):
    """qW-Map 0.1.2 ``qw_map.tanh``: pi * tanh(w)  (K13; from the published
    package -- not available offline)."""
    return math.pi * torch.tanh(w)


def post_process_dense(p, pixels: int):
    """``_post_process`` of the dense nets (``nn/qdense.py:49-54, 443-448``)."""
    return torch.clamp(p[:, :pixels] * pixels, 0, 1)


def qdense_undirected_forward(x_img, weights, shape, weight_map="qw_tanh"):
    """``QDenseUndirected_old.forward`` (``nn/qdense.py:56-62``, map at ``:45``)
    or ``QDenseUndirected_old_noise.forward`` (``:113-119``, ``torch.tanh`` at
    ``:97``) when ``weight_map == "tanh"``."""
    wd, ht = shape
    pixels = wd * ht
    n = math.ceil(math.log2(pixels))
    x = x_img.reshape(x_img.shape[0], -1).to(sv.RDT)
    w = weights.to(sv.RDT)
    w = qw_map_tanh(w) if weight_map == "qw_tanh" else torch.tanh(w)
    spec = Spec(n=n, encoding="amplitude", imprimitive="CNOT", measure="probs", pad_with=0.1)
    p = run_round(spec, x, w.unsqueeze(0))
    return post_process_dense(p, pixels).reshape(-1, 1, wd, ht)


def qnn_forward(x_img, lin_down_w, lin_down_b, weights, lin_up_w, lin_up_b):
    """``QNN_noise.forward`` / ``QNN.forward`` (``nn/qdense.py:267-289, 346-368``)."""
    b = x_img.shape[0]
    n = weights.shape[1]
    x = x_img.reshape(b, -1).to(sv.RDT)
    xr = x @ lin_down_w.to(sv.RDT).T + lin_down_b.to(sv.RDT)
    spec = Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz")
    ev = run_round(spec, xr, weights.to(sv.RDT).unsqueeze(0))
    out = ev @ lin_up_w.to(sv.RDT).T + lin_up_b.to(sv.RDT)
    return out.reshape(x_img.shape)


def qiddm_ll_forward(x_img, lin_down_w, lin_down_b, weights1, lin_up_w, lin_up_b):
    """``QIDDM_LL_noise.forward`` (``nn/qdense.py:1620-1642``); with PCA output in
    place of ``linear_down`` it is ``QIDDM_PL_noise.forward`` (``:1424-1448``)."""
    b = x_img.shape[0]
    n = weights1.shape[3]
    x = x_img.reshape(b, -1).to(sv.RDT)
    xr = x @ lin_down_w.to(sv.RDT).T + lin_down_b.to(sv.RDT)
    spec = Spec(n=n, encoding="rz", imprimitive="CZ", measure="expz")
    ev = run_circuit(spec, xr, weights1.to(sv.RDT))
    out = ev @ lin_up_w.to(sv.RDT).T + lin_up_b.to(sv.RDT)
    return out.reshape(x_img.shape)


def differn_from_reduced(x_reduced, weights, shape):
    """``differN_noise.forward`` from the post-PCA tensor on
    (``nn/qdense.py:458-472``; the PCA itself is host-side, finding F4)."""
    wd, ht = shape
    pixels = wd * ht
    n = weights.shape[3]
    spec = Spec(n=n, encoding="rz", imprimitive="CZ", measure="probs")
    # the reference casts the PCA output to float32 before the circuit (:458)
    x = x_reduced.to(torch.float32).to(sv.RDT)
    p = run_circuit(spec, x, weights.to(sv.RDT))
    return post_process_dense(p, pixels).reshape(-1, 1, wd, ht)


def qconv_wires(in_channels: int, out_channels: int, kernel_size) -> int:
    """``nn/qconv.py:24-28``."""
    kh, kw = kernel_size
    return max(math.ceil(math.log2(kh * kw * in_channels)), math.ceil(math.log2(out_channels)), 1)


def qconv2d_forward(x, weights, out_channels, kernel_size=(3, 3), padding=(1, 1)):
    """The *intended* ``_QConv2d_FAST.forward`` (finding F3): the reference body
    ``nn/qconv.py:71-87`` with ``x = self.qnode(x)`` restored between ``:78`` and
    ``:79``; circuit ``:51-56``; ``_post_process`` ``:58-69``."""
    b, c, h_in, w_in = x.shape
    kh, kw = kernel_size
    n = qconv_wires(c, out_channels, kernel_size)
    h_out = h_in + 2 * padding[0] - kh + 1
    w_out = w_in + 2 * padding[1] - kw + 1
    cols = torch.nn.functional.unfold(x.to(sv.RDT), kernel_size=kernel_size, padding=padding)
    feats = cols.permute(0, 2, 1).reshape(b * h_out * w_out, c * kh * kw)
    spec = Spec(n=n, encoding="amplitude", imprimitive="CNOT", measure="probs",
                pad_with=0.5, enc_offset=0.1)
    p = run_round(spec, feats, qw_map_tanh(weights.to(sv.RDT)).unsqueeze(0))
    p = torch.clamp(p * p.shape[-1] * 0.5, 0.0, 1.0)[:, ::2][:, :out_channels]
    return p.reshape(b, h_out, w_out, -1).permute(0, 3, 1, 2).contiguous()
