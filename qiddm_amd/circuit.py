"""Circuit descriptor + torch-level execution on the HIP statevector engine.

This is the layer directly above the C ABI: tensors in, tensors out, autograd
wired to the parameter-shift sweep.  It is what a QNode call resolves to
(``qiddm_amd/qml.py``) exactly where the reference calls PennyLane
(``self.qnode(inputs, weights)``, reference nn/qdense.py:58, 279, 465, 1633).

No CPU path: tensors must live on a HIP device and the in-tree extension must
be built, otherwise the call raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, replace

import torch

from . import _capi

_ENC = {"none": _capi.ENC_NONE, "amplitude": _capi.ENC_AMPLITUDE, "rz": _capi.ENC_RZ, "ry": _capi.ENC_RY,
        "ry_blocks": _capi.ENC_RY_BLOCKS}
_IMP = {"CNOT": _capi.IMP_CNOT, "CZ": _capi.IMP_CZ}
_MEAS = {"probs": _capi.MEAS_PROBS, "expz": _capi.MEAS_EXPZ}
_DT = {"f32": (_capi.F32, torch.float32), "f64": (_capi.F64, torch.float64)}

_default_precision = "f32"


def set_default_precision(p: str) -> None:
    """"f32" (complex64 state; production) or "f64" (complex128; the reference's precision)."""
    global _default_precision
    if p not in _DT:
        raise ValueError(f"precision must be 'f32' or 'f64', got {p!r}")
    _default_precision = p


def get_default_precision() -> str:
    return _default_precision


@dataclass(frozen=True)
class Circuit:
    """Host mirror of ``qiddm_circuit_t`` (see include/qiddm_hip.h for the family)."""

    n_qubits: int
    encoding: str = "rz"       # "none" | "amplitude" | "rz" | "ry" (once) | "ry_blocks" (every block)
    imprimitive: str = "CZ"    # "CNOT" | "CZ"
    measure: str = "expz"      # "probs" | "expz"
    n_rounds: int = 1
    n_blocks: int = 1
    sel_layers: int = 1
    n_features: int = 0        # 0 -> n_qubits for angle encodings
    enc_scale: float = 1.0
    enc_offset: float = 0.0
    pad_with: float = 0.0

    def __post_init__(self):
        for name, table in (("encoding", _ENC), ("imprimitive", _IMP), ("measure", _MEAS)):
            if getattr(self, name) not in table:
                raise ValueError(f"{name}={getattr(self, name)!r} not in {sorted(table)}")

    @property
    def dim(self) -> int:
        return 1 << self.n_qubits

    @property
    def out_cols(self) -> int:
        return self.dim if self.measure == "probs" else self.n_qubits

    @property
    def angles_shape(self):
        return (self.n_rounds, self.n_blocks, self.sel_layers, self.n_qubits, 3)

    @property
    def features(self) -> int:
        if self.encoding in ("rz", "ry", "ry_blocks"):
            return self.n_features or self.n_qubits
        return self.n_features

    def c_struct(self, precision: str) -> _capi.CircuitStruct:
        return _capi.CircuitStruct(
            n_qubits=self.n_qubits, encoding=_ENC[self.encoding], imprimitive=_IMP[self.imprimitive],
            measure=_MEAS[self.measure], n_rounds=self.n_rounds, n_blocks=self.n_blocks,
            sel_layers=self.sel_layers, n_features=self.features, dtype=_DT[precision][0], reserved=0,
            enc_scale=float(self.enc_scale), enc_offset=float(self.enc_offset),
            pad_with=float(self.pad_with))

    def gate_count(self) -> int:
        """Gate applications per sample per forward (SURVEY.md section 8a counting)."""
        n = self.n_qubits
        per_block = self.sel_layers * (n + (n if n > 1 else 0))
        if self.encoding == "rz":
            per_block += n
        g = self.n_rounds * self.n_blocks * per_block
        if self.encoding == "ry":
            g += self.n_rounds * n
        if self.encoding == "ry_blocks":
            g += self.n_rounds * self.n_blocks * n
        if self.encoding == "amplitude":
            g += self.n_rounds
        return g

    def algorithmic_bytes_per_sample(self, precision: str = "f32") -> float:
        """(G + 1/2) * 2 * 2^n * sizeof(complex)."""
        csize = 8 if precision == "f32" else 16
        return (self.gate_count() + 0.5) * 2 * self.dim * csize


def wide_sweeps_per_sample(circ: "Circuit", precision: str = "f32"):
    """(slab sweeps per sample, kernel name) of the n = 11..16 forward: one sweep = one read + one write of all 2^n
    amplitudes.  CZ circuits with no / RZ encoding run on ``wide_cz_kernel``; the others mirror the pass program the
    generic tiled kernel builds (``qsim_tiled.h: ProgramBuilder``): every layer of
    single-qubit gates costs one switch of the local-bit set, a CNOT ring closes one more pass; the first pass of a
    round only writes and the last one only reads, i.e. P passes = P - 1 sweeps."""
    if circ.n_qubits <= 10:
        return 0, "qiddm::circuit_kernel"
    t = "float" if precision == "f32" else "double"
    layers = circ.n_blocks * circ.sel_layers
    if circ.imprimitive == "CZ" and circ.encoding in ("none", "rz"):
        # qsim_wide_cz.h: a round's first layer is generated in registers, every further layer is one sweep
        return circ.n_rounds * (layers - 1), f"qiddm::wide_cz_kernel<{t}, {circ.n_qubits}>"
    passes = 1 + layers
    if circ.imprimitive == "CNOT":
        passes += layers
    if circ.encoding in ("ry", "ry_blocks"):
        passes += 1 if circ.encoding == "ry" else circ.n_blocks
    return circ.n_rounds * (passes - 1), f"qiddm::tiled_circuit_kernel<{t}, false>"


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def _require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"qiddm_amd: {what} lives on {t.device}; the quantum layers run only on a HIP device "
            "(MI355X). There is no CPU fallback -- move the module and its inputs to 'cuda'.")


def _prep_inputs(circ: Circuit, inputs, dtype, device):
    if circ.encoding == "none":
        return None, 0, None
    if inputs is None:
        raise ValueError(f"encoding {circ.encoding!r} needs inputs")
    x = inputs
    if x.dim() == 1:
        x = x.unsqueeze(0)
    if x.dim() != 2:
        raise ValueError(f"inputs must be (batch, features); got {tuple(inputs.shape)}")
    if circ.encoding == "amplitude":
        if x.shape[1] > circ.dim:
            # same message PennyLane raises from AmplitudeEmbedding
            raise ValueError(f"Features must be of length {circ.dim} or smaller; got length {x.shape[1]}.")
        if circ.features != x.shape[1]:
            circ = replace(circ, n_features=x.shape[1])
    elif x.shape[1] < circ.n_qubits:
        raise ValueError(f"angle encoding on {circ.n_qubits} wires needs >= {circ.n_qubits} "
                         f"feature columns; got {x.shape[1]}")
    x = x.to(device=device, dtype=dtype).contiguous()
    return x, x.shape[1], circ


_workspaces = {}


def _scratch(cache: dict, tag, need: int, device, floor: int = 0) -> torch.Tensor:
    """Launch scratch of >= ``need`` bytes on ``device`` for the current stream.

    Eager launches share one cached buffer per (tag, device, stream), grown on demand (stream order makes the
    reuse safe).  While the stream is being CAPTURED into a HIP graph the buffer is a fresh allocation from the
    graph's own memory pool instead -- owned by the recording, never cached, never replaced -- so a later eager call
    or another recording on a recycled stream handle cannot take a recorded pointer away."""
    size = max(int(need), int(floor), 1)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(size, dtype=torch.uint8, device=device)
    key = (tag, device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = cache.get(key)
    if buf is None or buf.numel() < size:
        buf = cache[key] = torch.empty(size, dtype=torch.uint8, device=device)
    return buf


def _workspace(circ: Circuit, precision: str, batch: int, n_replicas: int, device):
    """Per-device scratch for the n > 10 tiled kernel (slabs of concurrently resident workgroups).
    Returns (tensor, ptr, nbytes); (None, 0, 0) when the circuit is register-resident.  The caller keeps the TENSOR
    alive across its launch: during a graph capture it is a fresh allocation that nothing else references."""
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    need = lib.qiddm_workspace_bytes(ctypes.byref(cs), batch, n_replicas)
    if need < 0:
        _capi.check(-1)
    if need == 0:
        return None, 0, 0
    buf = _scratch(_workspaces, "tiled", need, device)
    return buf, buf.data_ptr(), need


def prepare_gates(circ: Circuit, angles: torch.Tensor, precision: str) -> torch.Tensor:
    """angles (N,L,S,n,3) -> gate table (G,7,8) of the compute dtype, on device."""
    _require_device(angles, "the circuit weights")
    if tuple(angles.shape) != circ.angles_shape:
        raise ValueError(f"angles must have shape {circ.angles_shape}; got {tuple(angles.shape)}")
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    a64 = angles.detach().to(torch.float64).contiguous()
    n_elems = lib.qiddm_gate_table_elems(ctypes.byref(cs))
    if n_elems < 0:
        _capi.check(-1)
    table = torch.empty(n_elems, dtype=_DT[precision][1], device=angles.device)
    _capi.check(lib.qiddm_prepare_gates(ctypes.byref(cs), a64.data_ptr(), table.data_ptr(),
                                        _stream_ptr(angles.device)))
    return table


def run_forward(circ: Circuit, inputs, angles: torch.Tensor, precision: str | None = None,
                table: torch.Tensor | None = None, batch: int | None = None) -> torch.Tensor:
    """Raw forward (no autograd).  Returns (B, 2^n) probabilities or (B, n) <Z>.  A circuit without a data encoding
    (``encoding="none"``: ``inputs`` is None) takes its row count from ``batch``."""
    precision = precision or _default_precision
    dtype = _DT[precision][1]
    _require_device(angles, "the circuit weights")
    device = angles.device
    x, ld, circ2 = _prep_inputs(circ, inputs, dtype, device)
    circ = circ2 or circ
    if x is None and batch is None:
        raise ValueError("run_forward needs inputs (or batch=) to know the batch size")
    batch = x.shape[0] if x is not None else int(batch)
    lib = _capi.lib()
    if table is None:
        table = prepare_gates(circ, angles, precision)
    out = torch.empty(batch, circ.out_cols, dtype=dtype, device=device)
    cs = circ.c_struct(precision)
    ws_buf, ws_ptr, ws_bytes = _workspace(circ, precision, batch, 0, device)
    _capi.check(lib.qiddm_forward(ctypes.byref(cs), 0 if x is None else x.data_ptr(), batch, ld, table.data_ptr(),
                                  out.data_ptr(), circ.out_cols, ws_ptr, ws_bytes, _stream_ptr(device)))
    return out


def run_forward_post(circ: Circuit, inputs, angles: torch.Tensor, post_cols: int, post_scale: float,
                     precision: str | None = None, table: torch.Tensor | None = None) -> torch.Tensor:
    """Probabilities of an n <= 10 circuit with the probability nets' ``_post_process`` fused into the store
    (``qiddm_forward_post``): ``clamp(p[:, :post_cols] * post_scale, 0, 1)`` as float64, (B, post_cols).  No autograd."""
    precision = precision or _default_precision
    dtype = _DT[precision][1]
    _require_device(angles, "the circuit weights")
    device = angles.device
    if circ.measure != "probs" or circ.n_qubits > 10:
        raise ValueError("run_forward_post: probabilities of an n <= 10 circuit")
    x, ld, circ2 = _prep_inputs(circ, inputs, dtype, device)
    circ = circ2 or circ
    if x is None:
        raise ValueError("run_forward_post needs inputs")
    if table is None:
        table = prepare_gates(circ, angles, precision)
    out = torch.empty(x.shape[0], post_cols, dtype=torch.float64, device=device)
    cs = circ.c_struct(precision)
    _capi.check(_capi.lib().qiddm_forward_post(ctypes.byref(cs), x.data_ptr(), x.shape[0], ld, table.data_ptr(),
                                               out.data_ptr(), out.stride(0), int(post_cols), float(post_scale),
                                               _stream_ptr(device)))
    return out


def _as_f64(t, device):
    return None if t is None else t.detach().to(device=device, dtype=torch.float64).contiguous()


def dense_forward(circ: Circuit, x: torch.Tensor, w_down, b_down, angles, w_up, b_up,
                  precision: str | None = None, post_mode: int = 0, noise_factor: float = 1.0):
    """linear_down -> chained circuit rounds -> linear_up (+ optional sampling update) in ONE
    launch (``qiddm_dense_forward``).  No autograd: inference / sampling path.
    x: (batch, in_features) -> (batch, out_features) float64."""
    precision = precision or _default_precision
    _require_device(angles, "the circuit weights")
    _require_device(x, "the input batch")
    device = angles.device
    if circ.encoding != "rz" or circ.measure != "expz":
        raise ValueError("dense_forward needs encoding='rz' and measure='expz'")
    if tuple(angles.shape) != circ.angles_shape:
        raise ValueError(f"angles must have shape {circ.angles_shape}; got {tuple(angles.shape)}")
    n = circ.n_qubits
    xx = _as_f64(x, device)
    wd, bd, wu, bu, ang = (_as_f64(t, device) for t in (w_down, b_down, w_up, b_up, angles))
    if wd.shape != (n, xx.shape[1]) or wu.shape[1] != n:
        raise ValueError(f"linear shapes {tuple(wd.shape)} / {tuple(wu.shape)} do not match n={n}, "
                         f"in_features={xx.shape[1]}")
    y = torch.empty(xx.shape[0], wu.shape[0], dtype=torch.float64, device=device)
    cs = circ.c_struct(precision)
    _capi.check(_capi.lib().qiddm_dense_forward(
        ctypes.byref(cs), xx.data_ptr(), xx.shape[0], xx.stride(0), xx.shape[1], wd.data_ptr(),
        0 if bd is None else bd.data_ptr(), ang.data_ptr(), wu.data_ptr(),
        0 if bu is None else bu.data_ptr(), wu.shape[0], int(post_mode), float(noise_factor),
        y.data_ptr(), y.stride(0), _stream_ptr(device)))
    return y


def dense_sample_tables(circ: Circuit, angles: torch.Tensor, precision: str | None = None) -> torch.Tensor:
    """The sampler's per-layer tables for these weights (``qiddm_dense_sample_prepare``): build once, pass to
    every ``dense_sample`` call until the angles change.  Raises ``QiddmError`` (-2) outside the sampler's range."""
    precision = precision or _default_precision
    _require_device(angles, "the circuit weights")
    device = angles.device
    ang = _as_f64(angles.detach(), device)
    if tuple(ang.shape) != circ.angles_shape:
        raise ValueError(f"angles must have shape {circ.angles_shape}; got {tuple(ang.shape)}")
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    need = lib.qiddm_dense_sample_tables_bytes(ctypes.byref(cs))
    if need < 0:
        _capi.check(int(need))
    tables = torch.empty(need, dtype=torch.uint8, device=device)
    _capi.check(lib.qiddm_dense_sample_prepare(ctypes.byref(cs), ang.data_ptr(), tables.data_ptr(),
                                               _stream_ptr(device)))
    return tables


def dense_sample(circ: Circuit, x: torch.Tensor, w_down, b_down, angles, w_up, b_up, n_steps: int,
                 precision: str | None = None, post_mode: int = 0, noise_factor: float = 1.0,
                 tables: torch.Tensor | None = None):
    """``n_steps`` consecutive bodies of the sampling loop in ONE launch (``qiddm_dense_sample``).
    Returns (n_steps, batch, features) float64 -- the image after every step.  Raises ``QiddmError``
    (status -2) when the configuration is outside the fused sampler's range."""
    precision = precision or _default_precision
    _require_device(angles, "the circuit weights")
    _require_device(x, "the input batch")
    device = angles.device
    xx = _as_f64(x, device)
    wd, bd, wu, bu, ang = (_as_f64(t, device) for t in (w_down, b_down, w_up, b_up, angles))
    if tuple(ang.shape) != circ.angles_shape:
        raise ValueError(f"angles must have shape {circ.angles_shape}; got {tuple(ang.shape)}")
    y = torch.empty(n_steps, xx.shape[0], wu.shape[0], dtype=torch.float64, device=device)
    cs = circ.c_struct(precision)
    _capi.check(_capi.lib().qiddm_dense_sample(
        ctypes.byref(cs), xx.data_ptr(), xx.shape[0], xx.stride(0), xx.shape[1], wd.data_ptr(),
        0 if bd is None else bd.data_ptr(), ang.data_ptr(), wu.data_ptr(), 0 if bu is None else bu.data_ptr(),
        wu.shape[0], int(post_mode), float(noise_factor), int(n_steps), y.data_ptr(), y.stride(1),
        y.stride(0), 0 if tables is None else tables.data_ptr(), _stream_ptr(device)))
    return y


# QIDDM_NO_LEAN_SAMPLER=1: keep the general four-wavefront sampler for every net (kernel experiments: A/B on one box)
_LEAN_SAMPLER = os.environ.get("QIDDM_NO_LEAN_SAMPLER", "0") != "1"


def dense_sample_lean_tables(circ: Circuit, angles, w_down, b_down, w_up, b_up, precision: str | None = None):
    """Tables of the lean sampling loop of the 8- and 6-qubit dense nets (``qiddm_dense_sample_lean_prepare``: tangent-form
    layers + the n x n map ``W_down W_up`` of the next step's angles), or None when the circuit is outside that
    kernel's family, the weights are outside the tangent form's range (``qiddm_dense_sample_lean_check``), or a HIP
    graph is being captured (the check reads one number back: build the tables once before recording)."""
    n = circ.n_qubits
    if not _LEAN_SAMPLER or n not in (6, 8) or circ.encoding != "rz" or circ.imprimitive != "CZ" \
            or circ.measure != "expz" or torch.cuda.is_current_stream_capturing():
        return None
    precision = precision or _default_precision
    device = angles.device
    ang = _as_f64(angles.detach(), device)
    wd, bd, wu, bu = (_as_f64(t, device) for t in (w_down, b_down, w_up, b_up))
    if tuple(ang.shape) != circ.angles_shape or wd.shape[0] != n or wu.shape != (wd.shape[1], n) or wd.shape[1] > 2048:
        return None
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    need = lib.qiddm_dense_sample_lean_tables_bytes(ctypes.byref(cs))
    if need < 0:
        return None                                  # e.g. deep float64 circuits whose tables do not fit in LDS
    tables = torch.empty(need, dtype=torch.uint8, device=device)
    st = _stream_ptr(device)
    _capi.check(lib.qiddm_dense_sample_lean_prepare(ctypes.byref(cs), ang.data_ptr(), wd.data_ptr(),
                                                    0 if bd is None else bd.data_ptr(), wu.data_ptr(),
                                                    0 if bu is None else bu.data_ptr(), wd.shape[1], tables.data_ptr(), st))
    ok = lib.qiddm_dense_sample_lean_check(ctypes.byref(cs), tables.data_ptr(), st)
    if ok < 0:
        _capi.check(ok)
    return tables if ok == 1 else None


def dense_sample_lean(circ: Circuit, x: torch.Tensor, w_down, b_down, w_up, b_up, n_steps: int, tables: torch.Tensor,
                      precision: str | None = None, post_mode: int = 0, noise_factor: float = 1.0):
    """``n_steps`` bodies of the sampling loop in ONE launch of the lean kernel (``qiddm_dense_sample_lean``;
    ``post_mode`` / ``noise_factor`` as for ``dense_sample``).  Returns (n_steps, batch, features) float64."""
    precision = precision or _default_precision
    _require_device(x, "the input batch")
    device = tables.device
    xx = _as_f64(x, device)
    wd, bd, wu, bu = (_as_f64(t, device) for t in (w_down, b_down, w_up, b_up))
    y = torch.empty(n_steps, xx.shape[0], wu.shape[0], dtype=torch.float64, device=device)
    cs = circ.c_struct(precision)
    _capi.check(_capi.lib().qiddm_dense_sample_lean(
        ctypes.byref(cs), xx.data_ptr(), xx.shape[0], xx.stride(0), xx.shape[1], wd.data_ptr(),
        0 if bd is None else bd.data_ptr(), wu.data_ptr(), 0 if bu is None else bu.data_ptr(), int(post_mode),
        float(noise_factor), int(n_steps), y.data_ptr(), y.stride(1), y.stride(0), tables.data_ptr(),
        _stream_ptr(device)))
    return y


def circuit_unitary(angles: torch.Tensor, n_qubits: int, imprimitive: str = "CNOT",
                    precision: str = "f64") -> torch.Tensor:
    """``qml.matrix`` of ``StronglyEntanglingLayers(angles (S, n, 3), imprimitive)`` with
    ``wire_order=range(n)`` (reference nn/qconv.py:96-103): (D, D) complex128 on the device
    (``qiddm_circuit_unitary``; ``qiddm_circuit_unitary_wide`` for n = 11, 12, returned as a transposed view)."""
    _require_device(angles, "the circuit weights")
    device = angles.device
    ang = _as_f64(angles, device)
    circ = Circuit(n_qubits=n_qubits, encoding="none", imprimitive=imprimitive, measure="probs", n_rounds=1,
                   n_blocks=1, sel_layers=ang.shape[0])
    if tuple(ang.reshape(circ.angles_shape).shape) != circ.angles_shape:
        raise ValueError(f"angles must have shape (S, {n_qubits}, 3); got {tuple(angles.shape)}")
    d = 1 << n_qubits
    u = torch.empty(d, d, 2, dtype=torch.float64, device=device)
    cs = circ.c_struct(precision)
    if n_qubits > 10:
        # the wide kernel writes U^T (columns contiguous); hand back the transposed view: indexing is <k|U|j> as
        # below, and qconv_unitary_forward recognises the layout
        _capi.check(_capi.lib().qiddm_circuit_unitary_wide(ctypes.byref(cs), ang.data_ptr(), u.data_ptr(),
                                                           _stream_ptr(device)))
        return torch.view_as_complex(u).transpose(0, 1)
    _capi.check(_capi.lib().qiddm_circuit_unitary(ctypes.byref(cs), ang.data_ptr(), u.data_ptr(),
                                                  _stream_ptr(device)))
    return torch.view_as_complex(u)


def dense_unitary_operand(unitary: torch.Tensor, cols: int) -> torch.Tensor:
    """(2^n, 2 cols) float32 operand ``[Re U^T | Im U^T]`` restricted to the first ``cols`` outcomes -- built once per
    weights for ``dense_unitary_forward``."""
    rows = unitary[:cols]                                   # rows k < cols of <k|U|j>
    return torch.cat([rows.real.t(), rows.imag.t()], dim=1).to(torch.float32).contiguous()


def dense_unitary_forward(x: torch.Tensor, operand: torch.Tensor, n_qubits: int, cols: int, pad_with: float,
                          post_scale: float) -> torch.Tensor:
    """``clamp(probs[:, :cols] * post_scale, 0, 1)`` of ``AmplitudeEmbedding(x, normalize=True, pad_with) -> U`` for a
    whole batch as one float32 matrix product with the cached unitary (``qiddm_amp_embed_rows`` -> library GEMM ->
    ``qiddm_prob_post``).  x: (B, features) float64 on the device; returns (B, cols) float64.  No autograd."""
    _require_device(x, "the input batch")
    device = x.device
    xx = _as_f64(x, device)
    if xx.stride(1) != 1:
        xx = xx.contiguous()
    b, f = xx.shape
    d = 1 << n_qubits
    lib = _capi.lib()
    st = _stream_ptr(device)
    v = torch.empty(b, d, dtype=torch.float32, device=device)
    _capi.check(lib.qiddm_amp_embed_rows(xx.data_ptr(), b, xx.stride(0), f, n_qubits, float(pad_with), 0.0, v.data_ptr(), st))
    amps = torch.mm(v, operand)                              # (B, 2 cols): the plain library GEMM
    out = torch.empty(b, cols, dtype=torch.float64, device=device)
    _capi.check(lib.qiddm_prob_post(amps.data_ptr(), b, cols, float(post_scale), out.data_ptr(), st))
    return out


_qconv_workspaces = {}


def qconv_unitary_forward(x: torch.Tensor, unitary: torch.Tensor, n_qubits: int, out_channels: int, kernel_size,
                          padding, upsample2x: bool = False, batch_norm: torch.nn.BatchNorm2d | None = None,
                          packed: dict | None = None, packed_key=None) -> torch.Tensor:
    """The eval-mode QConv2d forward (reference nn/qconv.py:105-113 + :58-69): unfold -> +0.1 ->
    AmplitudeEmbedding(pad 0.5) -> QubitUnitary(unitary) -> probs -> post-processing, as one implicit-im2col GEMM
    on the f32 matrix cores (``qiddm_qconv_unitary_forward``).  x (B, C, H, W) -> (B, out_channels, Ho, Wo)
    float64; ``unitary`` (D, D) complex128 from ``circuit_unitary``.

    ``upsample2x``: convolve ``Upsample(scale_factor=2, mode="bilinear")(x)`` without materialising it;
    ``batch_norm``: apply this (eval-mode, running statistics) BatchNorm2d to the output in the epilogue.
    ``packed`` / ``packed_key``: a dict owned by the caller (an eval-mode layer) and a key that changes with the
    unitary: the packed GEMM operand is kept there, with the BatchNorm's tensor versions, and the pack launch is
    skipped while nothing changed."""
    _require_device(unitary, "the circuit unitary")
    _require_device(x, "the input batch")
    device = unitary.device
    b, c, h, w = x.shape
    kh, kw = kernel_size
    ph, pw = padding
    d = 1 << n_qubits
    if tuple(unitary.shape) != (d, d) or unitary.dtype != torch.complex128:
        raise ValueError(f"unitary must be ({d}, {d}) complex128; got {tuple(unitary.shape)} {unitary.dtype}")
    xx = _as_f64(x, device).contiguous()
    transposed = (not unitary.is_contiguous()) and unitary.transpose(0, 1).is_contiguous()
    ur = torch.view_as_real(unitary.transpose(0, 1) if transposed else unitary.contiguous())
    lib = _capi.lib()
    need = lib.qiddm_qconv_unitary_workspace_bytes(n_qubits, c, kh, kw, out_channels)
    if need < 0:
        _capi.check(int(need))
    reuse = False
    if packed is not None and packed_key is not None and not torch.cuda.is_current_stream_capturing():
        bn_key = None if batch_norm is None else tuple(
            (None if t is None else (t._version, t.data_ptr()))
            for t in (batch_norm.weight, batch_norm.bias, batch_norm.running_mean, batch_norm.running_var)) + (batch_norm.eps,)
        key = (packed_key, bn_key, n_qubits, c, kh, kw, out_channels, str(device))
        entry = packed.get("entry")
        if entry is not None and entry[0] == key:
            ws, reuse = entry[1], True
        else:
            ws = torch.empty(max(int(need), 1), dtype=torch.uint8, device=device)
            packed["entry"] = (key, ws)
    else:
        ws = _scratch(_qconv_workspaces, "qconv", need, device)
    he, we = (2 * h, 2 * w) if upsample2x else (h, w)
    ho, wo = he + 2 * ph - kh + 1, we + 2 * pw - kw + 1
    y = torch.empty(b, out_channels, max(ho, 0), max(wo, 0), dtype=torch.float64, device=device)
    bn_ref, keep = None, []
    if batch_norm is not None:
        if batch_norm.running_mean is None or batch_norm.running_var is None:
            raise ValueError("the fused BatchNorm epilogue needs running statistics (eval mode)")

        def f64(t):
            if t is None:
                return 0
            t = _as_f64(t.detach(), device).contiguous()
            keep.append(t)
            return t.data_ptr()

        bn_struct = _capi.BatchNormStruct(weight=f64(batch_norm.weight), bias=f64(batch_norm.bias),
                                          running_mean=f64(batch_norm.running_mean),
                                          running_var=f64(batch_norm.running_var), eps=float(batch_norm.eps))
        bn_ref = ctypes.byref(bn_struct)
    _capi.check(lib.qiddm_qconv_unitary_forward(n_qubits, 0 if reuse else ur.data_ptr(), xx.data_ptr(), b, c, h, w, kh, kw, ph, pw,
                                                out_channels, int(bool(upsample2x)), bn_ref, int(transposed), y.data_ptr(),
                                                ws.data_ptr(), ws.numel(), _stream_ptr(device)))
    return y


def conv1x1_forward(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """Classical float64 1x1 convolution (``qiddm_conv1x1_forward``): x (B, C_in, H, W), weight (C_out, C_in, 1, 1)."""
    _require_device(x, "the input batch")
    device = x.device
    xx = _as_f64(x, device).contiguous()
    b, c, h, w = xx.shape
    wt = _as_f64(weight.detach(), device).reshape(weight.shape[0], -1).contiguous()
    if wt.shape[1] != c:
        raise ValueError(f"weight expects {wt.shape[1]} input channels, x has {c}")
    bs = None if bias is None else _as_f64(bias.detach(), device).contiguous()
    y = torch.empty(b, wt.shape[0], h, w, dtype=torch.float64, device=device)
    _capi.check(_capi.lib().qiddm_conv1x1_forward(xx.data_ptr(), wt.data_ptr(), 0 if bs is None else bs.data_ptr(), b,
                                                  c, wt.shape[0], h * w, y.data_ptr(), _stream_ptr(device)))
    return y


class _Conv1x1HeadFunction(torch.autograd.Function):
    """The UNets' one-channel ``final_conv`` (reference nn/unet.py:160-166) for training: ``qiddm_conv1x1_forward``, and
    the whole backward -- grad_x, grad_weight, grad_bias -- in one pass (``qiddm_conv1x1_head_backward``)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = conv1x1_forward(x, weight, bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        device = x.device
        xx = _as_f64(x, device).contiguous()
        b, c, h, w = xx.shape
        g = _as_f64(gy, device).contiguous()
        wt = _as_f64(weight.detach(), device).reshape(-1).contiguous()
        lib = _capi.lib()
        gx = torch.empty_like(xx) if ctx.needs_input_grad[0] else None
        gw = torch.empty(c, dtype=torch.float64, device=device) if ctx.needs_input_grad[1] else None
        gb = torch.empty(1, dtype=torch.float64, device=device) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        part = torch.empty(lib.qiddm_conv1x1_head_partials(b, h * w), c + 1, dtype=torch.float64, device=device)

        def ptr(t):
            return 0 if t is None else t.data_ptr()

        _capi.check(lib.qiddm_conv1x1_head_backward(xx.data_ptr(), wt.data_ptr(), g.data_ptr(), b, c, h * w, ptr(gx),
                                                    ptr(gw), ptr(gb), part.data_ptr(), _stream_ptr(device)))
        return (None if gx is None else gx.to(x.dtype)), (None if gw is None else gw.reshape(weight.shape).to(weight.dtype)), \
            (None if gb is None else gb)


def conv1x1_head(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    """Differentiable float64 1x1 convolution to ONE channel with at most 32 input channels (the UNets' head)."""
    return _Conv1x1HeadFunction.apply(x, weight, bias)


_train_workspaces = {}


_SCHEDULES_F32 = {}


def _schedule_f32(schedule: torch.Tensor, device) -> torch.Tensor:
    """The noising schedule as the flat float32 device tensor the step reads.  The schedule of a (tau, decay) pair is
    one cached tensor (noise._schedule), so its converted copy is cached too: converting it per step is an 11-element
    launch in front of every training step (5 us of the recorded C2 step)."""
    if schedule.dtype == torch.float32 and schedule.device == torch.device(device) and schedule.is_contiguous():
        return schedule.reshape(-1)
    key = (schedule.data_ptr(), schedule._version, schedule.dtype, tuple(schedule.shape), str(device))
    hit = _SCHEDULES_F32.get(key)
    if hit is None or hit[0] is not schedule:
        if len(_SCHEDULES_F32) > 64:
            _SCHEDULES_F32.clear()
        hit = (schedule, schedule.to(device=device, dtype=torch.float32).reshape(-1).contiguous())
        _SCHEDULES_F32[key] = hit
    return hit[1]


def train_step(circ: Circuit, x: torch.Tensor, noise: torch.Tensor, schedule: torch.Tensor, goal: str,
               w_down, b_down, angles, w_up, b_up, train_quantum: bool, want_recon: bool = False,
               want_elem_loss: bool = False, precision: str | None = None, rng_state: torch.Tensor | None = None):
    """One fused training step of the denoise loop (``qiddm_train_step``): noising, net forward, MSE and the
    backward pass of ``Diffusion.run_training_step_data/_noise`` (reference src/models.py:44-104) for a
    linear_down -> circuit -> linear_up net, in three launches and without a (batch*tau, pixels) tensor.

    x (B, P) float64; noise (B, P) float32; schedule (tau+1,) float32 with schedule[0] == 0.
    ``rng_state`` (2,) int64 on the device = {seed, offset}: the noise field is then generated inside the launch
    (Philox) and WRITTEN to ``noise``, and the offset is advanced -- no separate RNG launches.
    Returns a dict: ``loss`` (0-d), ``w_up``, ``b_up`` and -- when ``train_quantum`` -- ``w_down``, ``b_down``,
    ``angles`` gradients (float64, parameter-shaped), plus ``recon`` / ``elem_loss`` (B*tau, P) on request."""
    precision = precision or _default_precision
    _require_device(angles, "the circuit weights")
    _require_device(x, "the input batch")
    device = angles.device
    xx = _as_f64(x, device)
    if xx.stride(1) != 1:
        xx = xx.contiguous()
    nz = noise.to(device=device, dtype=torch.float32)
    if nz.stride(-1) != 1:
        nz = nz.contiguous()
    if rng_state is not None:
        if nz.data_ptr() != noise.data_ptr():
            raise ValueError("with rng_state the noise buffer is an output: pass a float32 device tensor")
        if rng_state.dtype != torch.int64 or rng_state.numel() != 2 or not rng_state.is_cuda:
            raise ValueError("rng_state must be a (2,) int64 device tensor {seed, offset}")
    sch = _schedule_f32(schedule, device)
    wd, bd, wu, bu, ang = (_as_f64(t, device) for t in (w_down, b_down, w_up, b_up, angles))
    if tuple(ang.shape) != circ.angles_shape:
        raise ValueError(f"angles must have shape {circ.angles_shape}; got {tuple(ang.shape)}")
    batch, pixels = xx.shape
    tau = sch.numel() - 1
    if tuple(nz.shape) != (batch, pixels):
        raise ValueError(f"noise must have shape {(batch, pixels)}; got {tuple(nz.shape)}")
    if tuple(wd.shape) != (circ.n_qubits, pixels) or tuple(wu.shape) != (pixels, circ.n_qubits):
        raise ValueError("linear_down / linear_up must map pixels -> n_qubits -> pixels")
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    need = lib.qiddm_train_workspace_bytes(ctypes.byref(cs), batch, pixels, tau)
    if need < 0:
        _capi.check(int(need))
    ws = _scratch(_train_workspaces, "train", need, device)
    f64 = dict(dtype=torch.float64, device=device)
    out = {"loss": torch.empty((), **f64), "w_up": torch.empty_like(wu), "b_up": torch.empty(pixels, **f64)}
    if train_quantum:
        out.update(w_down=torch.empty_like(wd), b_down=torch.empty(circ.n_qubits, **f64),
                   angles=torch.empty_like(ang))
    if want_recon:
        out["recon"] = torch.empty(batch * tau, pixels, **f64)
    if want_elem_loss:
        out["elem_loss"] = torch.empty(batch * tau, pixels, **f64)

    def ptr(t):
        return 0 if t is None else t.data_ptr()

    args = _capi.TrainArgs(
        x=ptr(xx), noise=ptr(nz), schedule=ptr(sch), x_ld=xx.stride(0), noise_ld=nz.stride(0), batch=batch,
        pixels=pixels, tau=tau, goal={"data": 0, "noise": 1}[goal], train_quantum=int(bool(train_quantum)),
        w_down=ptr(wd), b_down=ptr(bd), angles=ptr(ang), w_up=ptr(wu), b_up=ptr(bu), loss=ptr(out["loss"]),
        g_w_down=ptr(out.get("w_down")), g_b_down=ptr(out.get("b_down")), g_angles=ptr(out.get("angles")),
        g_w_up=ptr(out["w_up"]), g_b_up=ptr(out["b_up"]), recon=ptr(out.get("recon")),
        elem_loss=ptr(out.get("elem_loss")), rng_state=ptr(rng_state))
    _capi.check(lib.qiddm_train_step(ctypes.byref(cs), ctypes.byref(args), ws.data_ptr(), ws.numel(),
                                     _stream_ptr(device)))
    return out


def qconv_forward(x: torch.Tensor, angles: torch.Tensor, n_qubits: int, out_channels: int, kernel_size,
                  padding, precision: str | None = None) -> torch.Tensor:
    """The intended QConv2d forward in one launch (``qiddm_qconv_forward``); no autograd.
    x: (B, C, H, W) -> (B, out_channels, H_out, W_out) float64.  angles: (S, n, 3) after the
    pi*tanh map."""
    precision = precision or _default_precision
    _require_device(angles, "the circuit weights")
    _require_device(x, "the input batch")
    device = angles.device
    b, c, h, w = x.shape
    kh, kw = kernel_size
    ph, pw = padding
    circ = Circuit(n_qubits=n_qubits, encoding="amplitude", imprimitive="CNOT", measure="probs",
                   n_rounds=1, n_blocks=1, sel_layers=angles.shape[0], n_features=c * kh * kw,
                   enc_offset=0.1, pad_with=0.5)
    xx = _as_f64(x, device)
    ang = _as_f64(angles, device)
    ho, wo = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    y = torch.empty(b, out_channels, ho, wo, dtype=torch.float64, device=device)
    cs = circ.c_struct(precision)
    _capi.check(_capi.lib().qiddm_qconv_forward(ctypes.byref(cs), xx.data_ptr(), b, c, h, w, kh, kw, ph, pw,
                                                ang.data_ptr(), out_channels, y.data_ptr(),
                                                _stream_ptr(device)))
    return y


def _qconv_circuit(n_qubits, sel_layers, features):
    return Circuit(n_qubits=n_qubits, encoding="amplitude", imprimitive="CNOT", measure="probs",
                   n_rounds=1, n_blocks=1, sel_layers=sel_layers, n_features=features,
                   enc_offset=0.1, pad_with=0.5)


def qconv_backward(x: torch.Tensor, angles: torch.Tensor, grad_y: torch.Tensor, n_qubits: int, kernel_size, padding,
                   precision: str | None = None, with_input: bool = True):
    """Gradients of :func:`qconv_forward` (``qiddm_qconv_backward`` + ``qiddm_adjoint_finalize``).
    Returns (grad_angles (S, n, 3) float64, grad_x like x float64 or None)."""
    precision = precision or _default_precision
    dtype = _DT[precision][1]
    device = angles.device
    b, c, h, w = x.shape
    kh, kw = kernel_size
    ph, pw = padding
    out_channels = grad_y.shape[1]
    circ = _qconv_circuit(n_qubits, angles.shape[0], c * kh * kw)
    xx = _as_f64(x, device)
    gy = _as_f64(grad_y, device)
    ang = _as_f64(angles, device)
    ho, wo = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    m = b * ho * wo
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    table = prepare_gates(circ, ang.reshape(circ.angles_shape), precision)
    n_rot = lib.qiddm_num_rot_gates(ctypes.byref(cs))
    n_part = lib.qiddm_adjoint_partials(ctypes.byref(cs), m)
    if n_part < 0:
        _capi.check(-2)
    kp = torch.empty(n_part, n_rot, 8, dtype=dtype, device=device)
    gfeat = gx = None
    if with_input:
        gfeat = torch.empty(m, c * kh * kw, dtype=dtype, device=device)
        gx = torch.empty(b, c, h, w, dtype=torch.float64, device=device)
    _capi.check(lib.qiddm_qconv_backward(ctypes.byref(cs), xx.data_ptr(), b, c, h, w, kh, kw, ph, pw,
                                         table.data_ptr(), gy.data_ptr(), out_channels, kp.data_ptr(),
                                         0 if gfeat is None else gfeat.data_ptr(),
                                         0 if gx is None else gx.data_ptr(), _stream_ptr(device)))
    ga = torch.empty(n_rot, 3, dtype=torch.float64, device=device)
    _capi.check(lib.qiddm_adjoint_finalize(ctypes.byref(cs), ang.data_ptr(), kp.data_ptr(), n_part,
                                           ga.data_ptr(), _stream_ptr(device)))
    return ga.reshape(angles.shape), gx


class _QConvFunction(torch.autograd.Function):
    """The intended QConv2d layer (reference nn/qconv.py:51-87, finding F3) as one differentiable op: forward is
    the fused convolution launch, backward the adjoint sweep over the output pixels + the fold onto the image."""

    @staticmethod
    def forward(ctx, x, angles, n_qubits, out_channels, kernel_size, padding, precision):
        ctx.save_for_backward(x, angles)
        ctx.cfg = (n_qubits, kernel_size, padding, precision)
        return qconv_forward(x, angles.detach(), n_qubits, out_channels, kernel_size, padding, precision)

    @staticmethod
    def backward(ctx, grad_y):
        x, angles = ctx.saved_tensors
        n_qubits, kernel_size, padding, precision = ctx.cfg
        ga, gx = qconv_backward(x, angles.detach(), grad_y, n_qubits, kernel_size, padding, precision,
                                with_input=ctx.needs_input_grad[0])
        return (None if gx is None else gx.to(x.dtype)), ga.to(angles.dtype), None, None, None, None, None


def qconv_execute(x: torch.Tensor, angles: torch.Tensor, n_qubits: int, out_channels: int, kernel_size, padding,
                  precision: str | None = None) -> torch.Tensor:
    """Differentiable :func:`qconv_forward` (n <= 10)."""
    return _QConvFunction.apply(x, angles, n_qubits, out_channels, tuple(kernel_size), tuple(padding),
                                precision or _default_precision)


def _row_channels(out_channels: int):
    for co in (8, 16, 32):
        if out_channels <= co:
            return co
    return None


def qconv_unitary_route(n_qubits: int, in_channels: int, kernel_size, out_channels: int):
    """Which backward serves a QConv2d trained through its circuit unitary: ``"thin"`` (the hand-written
    thin-product kernel, ``qiddm_qconv_train_backward``: <= 32 output channels), ``"gemm"`` (wider layers, e.g. C4's
    256 channels on 12 wires: the same three products as library GEMMs over batch chunks) or None."""
    f = in_channels * kernel_size[0] * kernel_size[1]
    if not 2 <= n_qubits <= 12 or 2 * out_channels > 2 ** n_qubits or f > 2 ** n_qubits:
        return None
    co = _row_channels(out_channels)
    if co is not None and max(kernel_size) <= 15 and f + 1 <= 512:
        v_stride = (f + 1) | 1
        lds = ((f + 1) * 2 * co + 64 * v_stride + 64 * (2 * co + 1) + 8 * 64 * (co + 1) + 16 * 64) * 4 + f * 4
        if lds <= 160 * 1024:
            return "thin"
    return "gemm"


def qconv_unitary_trainable(n_qubits: int, in_channels: int, kernel_size, out_channels: int) -> bool:
    return qconv_unitary_route(n_qubits, in_channels, kernel_size, out_channels) is not None


# The weight-gradient chain of a quantum convolution (start vectors, gate table, one adjoint sweep per output channel,
# finalize, then the pi * tanh backward of the weights: ~90 us of small launches per layer) hangs off the layer's
# backward but nothing downstream of the layer needs it before the optimizer step.  It runs on a side stream:
# the ANGLES are computed on that stream in the forward, so autograd schedules their backward nodes there by itself
# (a backward node runs on its forward's stream; the engine joins the leaf streams when backward() returns), and the
# Function's backward enqueues the chain there after the thin-product kernel.  Inside a HIP-graph recording the fork
# becomes a parallel branch of the graph.  QIDDM_NO_WEIGHT_GRAD_STREAM=1 keeps everything on one stream.
_WEIGHT_GRAD_STREAM = os.environ.get("QIDDM_NO_WEIGHT_GRAD_STREAM", "0") != "1"
_weight_grad_streams = {}


def weight_grad_stream(device):
    """The side stream of ``device`` for the convolutions' weight-gradient chains; None when switched off."""
    if not _WEIGHT_GRAD_STREAM or device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _weight_grad_streams.get(idx)
    if side is None:
        side = _weight_grad_streams[idx] = torch.cuda.Stream(device=idx)
        # a leaf's AccumulateGrad node that outlives an iteration (the previous loss still referenced) keeps the stream it
        # was created under; the engine synchronises the two streams itself, which is all this design needs -- the
        # warning about it would fire on every step
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
    return None if torch.cuda.current_stream(idx) == side else side


def on_weight_grad_stream(fn, weights: torch.Tensor) -> torch.Tensor:
    """``fn(weights)`` (the layer's angle map) enqueued on the weight-gradient stream when autograd is recording, so
    that its backward nodes run there; the current stream waits for the result."""
    side = weight_grad_stream(weights.device) if torch.is_grad_enabled() and weights.requires_grad else None
    if side is None:
        return fn(weights)
    main = torch.cuda.current_stream(weights.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn(weights)
    main.wait_stream(side)
    out.record_stream(main)
    return out


def _angle_grads_off_stream(hpart, n_part, angles, n_qubits, f, c_out, co, device):
    """``_angle_grads_from_h`` behind everything the current stream has enqueued, on the weight-gradient stream."""
    side = weight_grad_stream(device)
    if side is None:
        return _angle_grads_from_h(hpart, n_part, angles, n_qubits, f, c_out, co, device).to(angles.dtype)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        ga = _angle_grads_from_h(hpart, n_part, angles, n_qubits, f, c_out, co, device).to(angles.dtype)
    hpart.record_stream(side)
    return ga


def _unitary_rows(u, n_qubits, f, c_out, co, device):
    """(F + 1, 2 co) float32 rows table of the backward (``qiddm_qconv_train_rows``)."""
    transposed = (not u.is_contiguous()) and u.transpose(0, 1).is_contiguous()
    ur = torch.view_as_real(u.transpose(0, 1) if transposed else u.contiguous())
    rt = torch.empty(f + 1, 2 * co, dtype=torch.float32, device=device)
    _capi.check(_capi.lib().qiddm_qconv_train_rows(n_qubits, ur.data_ptr(), int(transposed), f, c_out, co,
                                                   rt.data_ptr(), _stream_ptr(device)))
    return rt


def _angle_grads_from_h(hpart, n_part, angles, n_qubits, f, c_out, co, device):
    """dL/dangles = 2 Re sum_c <e_2c| dU/dangle |h_c> from the h partial slabs: start vectors
    (``qiddm_qconv_train_vectors``), one adjoint sweep per channel (``qiddm_matrix_adjoint``), finalize."""
    lib = _capi.lib()
    st = _stream_ptr(device)
    d = 1 << n_qubits
    psi0 = torch.empty(c_out, d, 2, dtype=torch.float64, device=device)
    lam = torch.empty(c_out, d, 2, dtype=torch.float64, device=device)
    _capi.check(lib.qiddm_qconv_train_vectors(n_qubits, hpart.data_ptr(), n_part, f, c_out, co, psi0.data_ptr(),
                                              lam.data_ptr(), st))
    ang = _as_f64(angles.detach(), device).contiguous()
    circ = Circuit(n_qubits=n_qubits, encoding="none", imprimitive="CNOT", measure="probs", n_rounds=1,
                   n_blocks=1, sel_layers=ang.shape[0])
    cs = circ.c_struct("f64")
    table = prepare_gates(circ, ang.reshape(circ.angles_shape), "f64")
    n_rot = lib.qiddm_num_rot_gates(ctypes.byref(cs))
    kparts = lib.qiddm_matrix_adjoint_partials(c_out)
    kp = torch.empty(kparts, n_rot, 8, dtype=torch.float64, device=device)
    need = lib.qiddm_matrix_adjoint_workspace_bytes(ctypes.byref(cs), c_out)
    ws = _scratch(_workspaces, "matrix-adjoint", need, device)
    _capi.check(lib.qiddm_matrix_adjoint(ctypes.byref(cs), psi0.data_ptr(), lam.data_ptr(), c_out, table.data_ptr(),
                                         kp.data_ptr(), ws.data_ptr(), ws.numel(), st))
    ga = torch.empty(n_rot, 3, dtype=torch.float64, device=device)
    _capi.check(lib.qiddm_adjoint_finalize(ctypes.byref(cs), ang.data_ptr(), kp.data_ptr(), kparts, ga.data_ptr(), st))
    return ga.reshape(angles.shape)


# QIDDM_QCONV_X32=1: hand the thin-product backward a float32 copy of the activations (qiddm_qconv_train_backward_x32).
# Measured at the unet_simple layer shapes (tools/stamp_qconv_train.py) the copy pass costs what the lighter gather gains
# (1.35 vs 1.26 ms, 0.95 vs 0.93 ms, 0.34 vs 0.36 ms per layer backward), so float64 activations go in as they are
_QCONV_X32 = os.environ.get("QIDDM_QCONV_X32", "0") == "1"
# QIDDM_QCONV_FOLD=1: keep the (F, M) feature gradients + fold for dL/dx instead of the per-pixel rows + transposed
# convolution (qiddm_qconv_train_backward_dx); the library reads the same variable
_QCONV_DX = os.environ.get("QIDDM_QCONV_FOLD") is None
# QIDDM_QCONV_BN_SPLIT=1: keep [QConv2d, BatchNorm2d] as two autograd nodes in training (no folding of the BatchNorm
# backward into the convolution's)
_QCONV_BN_FUSED = os.environ.get("QIDDM_QCONV_BN_SPLIT") is None
_GEMM_CHUNK_BYTES = 256 << 20      # patch matrix of one batch chunk on the "gemm" route


def _qconv_unitary_backward_gemm(x, grad_y, u, n_qubits, c_out, kernel_size, padding, need_gx):
    """The three products of the unitary-route backward as float32 library GEMMs (rocBLAS through torch) over batch
    chunks -- for layers beyond the thin-product kernel (C4: 2304 patch features x 512 columns).  Same formulas as
    qsim_qconv_train.h; the fold, the start vectors and the per-channel sweeps are the same HIP entry points."""
    from .nn.utils import unfold_patches
    device = x.device
    b, c, h, w = x.shape
    kh, kw = kernel_size
    ph, pw = padding
    f, d = c * kh * kw, 1 << n_qubits
    ho, wo = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    rt = _unitary_rows(u, n_qubits, f, c_out, c_out, device)
    r, rp = rt[:f], rt[f]                                   # (F, 2C), (2C)
    post = 0.5 * d
    hsum = torch.zeros(2 * c_out, f + 1, dtype=torch.float64, device=device)
    gx = torch.empty(b, c, h, w, dtype=torch.float64, device=device) if need_gx else None
    chunk = max(1, min(b, _GEMM_CHUNK_BYTES // max(ho * wo * f * 4, 1)))
    lib = _capi.lib()
    for b0 in range(0, b, chunk):
        xb = x[b0:b0 + chunk]
        cb = xb.shape[0]
        v = unfold_patches(xb.to(torch.float32), kernel_size, padding) + 0.1          # (Mc, F)
        inv = ((v * v).sum(dim=1, keepdim=True) + 0.25 * (d - f)).rsqrt()
        a = torch.addmm(rp, v, r) * inv                                                # (Mc, 2C): Re | Im
        p2 = a[:, :c_out] ** 2 + a[:, c_out:] ** 2
        g = grad_y[b0:b0 + cb].permute(0, 2, 3, 1).reshape(-1, c_out).to(torch.float32)
        t = torch.where(p2 * post <= 1.0, g * post, torch.zeros_like(g))
        dot = 2.0 * (t * p2).sum(dim=1, keepdim=True)
        w2 = torch.cat([t * a[:, :c_out], t * a[:, c_out:]], dim=1)                    # (Mc, 2C)
        v *= inv                                                                       # v^
        hsum[:, :f] += (w2.t() @ v).double()
        hsum[:, f] += (w2.t() @ (0.5 * inv)).double()[:, 0]
        if need_gx:
            gft = (2.0 * (r @ w2.t()) - v.t() * dot.t()) * inv.t()                     # (F, Mc), transposed gradients
            gft = gft.contiguous()
            _capi.check(lib.qiddm_qconv_fold_features(gft.data_ptr(), cb, c, h, w, kh, kw, ph, pw,
                                                      gx[b0:b0 + cb].data_ptr(), _stream_ptr(device)))
    return hsum.to(torch.float32).unsqueeze(0).contiguous(), gx


class _QConvUnitaryFunction(torch.autograd.Function):
    """QConv2d for training through the circuit's unitary: forward = unitary + the matrix-core GEMM of the eval
    route; backward = two more thin products per pixel tile and ONE adjoint sweep per output channel
    (``qiddm_qconv_train_backward`` + ``qiddm_matrix_adjoint``) instead of one sweep per output pixel."""

    @staticmethod
    def forward(ctx, x, angles, n_qubits, out_channels, kernel_size, padding):
        u = circuit_unitary(angles.detach(), n_qubits, "CNOT")
        y = qconv_unitary_forward(x, u, n_qubits, out_channels, kernel_size, padding)
        ctx.save_for_backward(x, angles, u)
        ctx.cfg = (n_qubits, out_channels, kernel_size, padding)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, angles, u = ctx.saved_tensors
        gx, ga = _qconv_unitary_backward(x, angles, u, ctx.cfg, grad_y, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, ga, None, None, None, None


def _qconv_unitary_backward(x, angles, u, cfg, grad_y, need_gx, need_ga, bn=None):
    """Backward of the unitary-route QConv2d: (dL/dx or None, dL/dangles or None).  ``bn = (conv_y, coef)``: grad_y is the
    gradient BEHIND the training-mode BatchNorm2d that follows the layer, and dL/dy is formed per channel as
    ``coef[0] * grad_y + coef[1] * conv_y + coef[2]`` inside the thin-product kernel (``qiddm_qconv_train_backward_bn``)."""
    n_qubits, c_out, (kh, kw), (ph, pw) = cfg
    device = x.device
    b, c, h, w = x.shape
    f = c * kh * kw
    ho, wo = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    if qconv_unitary_route(n_qubits, c, (kh, kw), c_out) == "gemm":
        if bn is not None:      # library-GEMM route: apply the BatchNorm coefficients with torch
            conv_y, coef = bn
            shape = (1, c_out, 1, 1)
            grad_y = coef[0].view(shape) * grad_y + coef[1].view(shape) * conv_y + coef[2].view(shape)
        hpart, gx = _qconv_unitary_backward_gemm(x, grad_y, u, n_qubits, c_out, (kh, kw), (ph, pw), need_gx)
        ga = _angle_grads_off_stream(hpart, 1, angles, n_qubits, f, c_out, c_out, device) if need_ga else None
        return (None if gx is None else gx.to(x.dtype)), ga
    co = _row_channels(c_out)
    lib = _capi.lib()
    st = _stream_ptr(device)
    rt = _unitary_rows(u, n_qubits, f, c_out, co, device)
    # the matrix-core kernel converts every patch element to float32 anyway: hand it a float32 copy (one elementwise
    # pass; its gather then holds half the bytes in flight)
    x32 = bn is None and _QCONV_X32 and bool(lib.qiddm_qconv_train_x32_ok(b, c, h, w, kh, kw, ph, pw, c_out, co))
    xx = x.detach().to(device=device, dtype=torch.float32).contiguous() if x32 else _as_f64(x, device).contiguous()
    n_part = lib.qiddm_qconv_train_partials(b, ho, wo, f)
    hpart = torch.empty(n_part, 2 * co, f + 1, dtype=torch.float32, device=device)
    gx = torch.empty(b, c, h, w, dtype=torch.float64, device=device) if need_gx else None
    # dL/dx from 2 co + 1 floats per pixel where the layer allows it (same-size convolution on the matrix-core
    # kernel), instead of the (F, M) feature gradients and their fold
    dx_elems = 0 if (gx is None or x32 or not _QCONV_DX) else \
        lib.qiddm_qconv_train_dx_elems(n_qubits, b, c, h, w, kh, kw, ph, pw, c_out, co)
    wpix = torch.empty(dx_elems, dtype=torch.float32, device=device) if dx_elems > 0 else None
    gfeat_t = torch.empty(f, b * ho * wo, dtype=torch.float32, device=device) if wpix is None else None
    # a channel slice of a wider contiguous gradient (one half of torch.cat's backward) is read in place by the
    # per-pixel-row entry point; anything else is made dense first
    gy = _as_f64(grad_y, device)
    gy_bstride = 0
    if wpix is not None and bn is None and not gy.is_contiguous() and gy.dim() == 4 and b > 1 \
            and gy.stride()[1:] == (ho * wo, wo, 1) and gy.stride(0) >= c_out * ho * wo \
            and b * gy.stride(0) < (1 << 32):
        gy_bstride = gy.stride(0)
    else:
        gy = gy.contiguous()

    def ptr(t):
        return 0 if t is None else t.data_ptr()

    if bn is not None:
        conv_y, coef = bn
        _capi.check(lib.qiddm_qconv_train_backward_bn(n_qubits, xx.data_ptr(), b, c, h, w, kh, kw, ph, pw, gy.data_ptr(),
                                                      conv_y.data_ptr(), coef.data_ptr(), c_out, rt.data_ptr(), co,
                                                      ptr(gfeat_t), ptr(wpix), hpart.data_ptr(), ptr(gx), st))
    elif wpix is not None:
        _capi.check(lib.qiddm_qconv_train_backward_dx(n_qubits, xx.data_ptr(), b, c, h, w, kh, kw, ph, pw,
                                                      gy.data_ptr(), gy_bstride, c_out, rt.data_ptr(), co,
                                                      wpix.data_ptr(), hpart.data_ptr(), gx.data_ptr(), st))
    else:
        entry = lib.qiddm_qconv_train_backward_x32 if x32 else lib.qiddm_qconv_train_backward
        _capi.check(entry(n_qubits, xx.data_ptr(), b, c, h, w, kh, kw, ph, pw, gy.data_ptr(), c_out, rt.data_ptr(),
                          co, gfeat_t.data_ptr(), hpart.data_ptr(), ptr(gx), st))
    ga = _angle_grads_off_stream(hpart, n_part, angles, n_qubits, f, c_out, co, device) if need_ga else None
    return (None if gx is None else gx.to(x.dtype)), ga


def qconv_unitary_execute(x: torch.Tensor, angles: torch.Tensor, n_qubits: int, out_channels: int, kernel_size,
                          padding) -> torch.Tensor:
    """Differentiable QConv2d through the circuit unitary (float32 products; see ``qconv_unitary_trainable``)."""
    return _QConvUnitaryFunction.apply(x, angles, n_qubits, out_channels, tuple(kernel_size), tuple(padding))


def _norm_workspace(batch, channels, hw, device):
    need = _capi.lib().qiddm_batchnorm_workspace_bytes(batch, channels, hw)
    if need < 0:
        _capi.check(-1)
    return _scratch(_workspaces, "norm", need, device, floor=1 << 16)


class _BatchNormTrainFunction(torch.autograd.Function):
    """Training-mode ``BatchNorm2d`` in float64 (``qiddm_batchnorm_train_forward`` / ``_backward``); moves the
    running statistics in place like the torch module."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        b, c = x.shape[:2]
        hw = x.numel() // max(b * c, 1)
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float64, device=x.device)
        invstd = torch.empty_like(mean)
        ws = _norm_workspace(b, c, hw, x.device)

        def ptr(t):
            return 0 if t is None else t.data_ptr()

        _capi.check(_capi.lib().qiddm_batchnorm_train_forward(
            x.data_ptr(), b, c, hw, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), float(momentum),
            float(eps), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(),
            _stream_ptr(x.device)))
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, invstd = ctx.saved_tensors
        b, c = x.shape[:2]
        hw = x.numel() // max(b * c, 1)
        gy = gy.contiguous()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw = torch.empty(c, dtype=torch.float64, device=x.device) if weight is not None else None
        gb = torch.empty(c, dtype=torch.float64, device=x.device) if ctx.has_bias else None
        ws = _norm_workspace(b, c, hw, x.device)

        def ptr(t):
            return 0 if t is None else t.data_ptr()

        _capi.check(_capi.lib().qiddm_batchnorm_backward(
            x.data_ptr(), gy.data_ptr(), b, c, hw, ptr(weight), mean.data_ptr(), invstd.data_ptr(), ptr(gx), ptr(gw),
            ptr(gb), ws.data_ptr(), ws.numel(), _stream_ptr(x.device)))
        return gx, gw, gb, None, None, None, None


def batch_norm_train(bn: torch.nn.BatchNorm2d, x: torch.Tensor) -> torch.Tensor:
    """``bn(x)`` for a float64 ``BatchNorm2d`` in training mode on the device, through the HIP kernels; anything else
    (eval mode, cumulative-average momentum, other dtypes, empty batches, CPU) goes to the torch module."""
    if not (bn.training and x.is_cuda and x.dtype == torch.float64 and x.dim() == 4 and x.numel() > 0
            and bn.momentum is not None and (bn.weight is None or bn.weight.dtype == torch.float64)
            and (bn.running_mean is None or bn.running_mean.dtype == torch.float64)):
        return bn(x)
    if x.shape[1] != bn.num_features:
        return bn(x)       # torch raises its own message
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BatchNormTrainFunction.apply(x, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps)


class _QConvBNTrainFunction(torch.autograd.Function):
    """``[QConv2d, BatchNorm2d]`` in training mode as ONE autograd node (every ``net`` of unet_simple, reference
    nn/unet_simple.py:9-18): the forward is the two layers' own launches; the backward runs the BatchNorm statistics
    pass only (``qiddm_batchnorm_backward_stats``) and lets the convolution's thin-product kernel apply the per-channel
    coefficients while it loads its dL/dy -- the BatchNorm backward's transform pass (two tensors read, one written) and
    the intermediate gradient tensor are gone.  The convolution's output never leaves the node, so nothing else can
    depend on its gradient."""

    @staticmethod
    def forward(ctx, x, angles, bn_weight, bn_bias, running_mean, running_var, momentum, eps, n_qubits, out_channels,
                kernel_size, padding):
        u = circuit_unitary(angles.detach(), n_qubits, "CNOT")
        y = qconv_unitary_forward(x, u, n_qubits, out_channels, kernel_size, padding)
        b, c = y.shape[:2]
        hw = y.numel() // max(b * c, 1)
        out = torch.empty_like(y)
        mean = torch.empty(c, dtype=torch.float64, device=y.device)
        invstd = torch.empty_like(mean)
        ws = _norm_workspace(b, c, hw, y.device)

        def ptr(t):
            return 0 if t is None else t.data_ptr()

        _capi.check(_capi.lib().qiddm_batchnorm_train_forward(
            y.data_ptr(), b, c, hw, ptr(bn_weight), ptr(bn_bias), ptr(running_mean), ptr(running_var), float(momentum),
            float(eps), out.data_ptr(), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), ws.numel(),
            _stream_ptr(y.device)))
        ctx.save_for_backward(x, angles, u, y, bn_weight, mean, invstd)
        ctx.cfg = (n_qubits, out_channels, kernel_size, padding)
        ctx.has_bias = bn_bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, angles, u, y, bn_weight, mean, invstd = ctx.saved_tensors
        b, c = y.shape[:2]
        hw = y.numel() // max(b * c, 1)
        device = y.device
        g = _as_f64(g, device).contiguous()
        gw = torch.empty(c, dtype=torch.float64, device=device) if bn_weight is not None else None
        gb = torch.empty(c, dtype=torch.float64, device=device) if ctx.has_bias else None
        coef = torch.empty(3, c, dtype=torch.float64, device=device)
        ws = _norm_workspace(b, c, hw, device)

        def ptr(t):
            return 0 if t is None else t.data_ptr()

        _capi.check(_capi.lib().qiddm_batchnorm_backward_stats(
            y.data_ptr(), g.data_ptr(), b, c, hw, ptr(bn_weight), mean.data_ptr(), invstd.data_ptr(), ptr(gw), ptr(gb),
            coef.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(device)))
        gx, ga = _qconv_unitary_backward(x, angles, u, ctx.cfg, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                         bn=(y, coef))
        return (gx, ga, gw, gb) + (None,) * 8


def qconv_bn_foldable(x_shape, n_qubits: int, out_channels: int, kernel_size, padding) -> bool:
    """Whether the unitary-route backward of this layer can apply a following BatchNorm's backward transform itself
    (``qiddm_qconv_train_bn_ok``; the library-GEMM route of the widest layers does it with torch)."""
    b, c, h, w = x_shape
    (kh, kw), (ph, pw) = kernel_size, padding
    route = qconv_unitary_route(n_qubits, c, (kh, kw), out_channels)
    if route != "thin":
        return route == "gemm"
    return bool(_capi.lib().qiddm_qconv_train_bn_ok(b, c, h, w, kh, kw, ph, pw, out_channels, _row_channels(out_channels)))


def batch_norm_eligible(bn: torch.nn.BatchNorm2d, channels: int) -> bool:
    """Whether a float64 training-mode ``BatchNorm2d`` runs on the HIP kernels (the module-side half of
    ``batch_norm_train``'s test)."""
    return bool(type(bn) is torch.nn.BatchNorm2d and bn.training and bn.momentum is not None
                and bn.num_features == channels
                and (bn.weight is None or bn.weight.dtype == torch.float64)
                and (bn.running_mean is None or bn.running_mean.dtype == torch.float64))


def qconv_bn_train(x: torch.Tensor, angles: torch.Tensor, bn: torch.nn.BatchNorm2d, n_qubits: int, out_channels: int,
                   kernel_size, padding) -> torch.Tensor:
    """``bn(QConv2d(x))`` through the circuit unitary with the BatchNorm backward folded into the convolution's
    (``_QConvBNTrainFunction``).  The caller has checked ``qconv_unitary_trainable`` and ``batch_norm_eligible``."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _QConvBNTrainFunction.apply(x, angles, bn.weight, bn.bias, rm, rv, bn.momentum, bn.eps, n_qubits,
                                       out_channels, tuple(kernel_size), tuple(padding))


class _Upsample2xFunction(torch.autograd.Function):
    """Bilinear x2 of a float64 (B, C, H, W) tensor with given 1-D interpolation matrices
    (``qiddm_upsample2x_forward`` / ``_backward``)."""

    @staticmethod
    def forward(ctx, x, a_h, a_w):
        x = x.contiguous()
        b, c, h, w = x.shape
        y = torch.empty(b, c, 2 * h, 2 * w, dtype=torch.float64, device=x.device)
        _capi.check(_capi.lib().qiddm_upsample2x_forward(x.data_ptr(), b * c, h, w, a_h.data_ptr(), a_w.data_ptr(),
                                                         y.data_ptr(), _stream_ptr(x.device)))
        ctx.save_for_backward(a_h, a_w)
        ctx.shape = (b, c, h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        a_h, a_w = ctx.saved_tensors
        b, c, h, w = ctx.shape
        gy = gy.contiguous()
        gx = torch.empty(b, c, h, w, dtype=torch.float64, device=gy.device)
        _capi.check(_capi.lib().qiddm_upsample2x_backward(gy.data_ptr(), b * c, h, w, a_h.data_ptr(), a_w.data_ptr(),
                                                          gx.data_ptr(), _stream_ptr(gy.device)))
        return gx, None, None


class _MaxPool2Function(torch.autograd.Function):
    """``MaxPool2d(2, 2)`` of a float64 (B, C, H, W) tensor (``qiddm_maxpool2_forward`` / ``_backward``): no index
    tensor, the backward finds the winners again from the saved input."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        b, c, h, w = x.shape
        y = torch.empty(b, c, h // 2, w // 2, dtype=torch.float64, device=x.device)
        _capi.check(_capi.lib().qiddm_maxpool2_forward(x.data_ptr(), b * c, h, w, y.data_ptr(), _stream_ptr(x.device)))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        b, c, h, w = x.shape
        gy = gy.contiguous()
        gx = torch.empty_like(x)
        _capi.check(_capi.lib().qiddm_maxpool2_backward(x.data_ptr(), gy.data_ptr(), b * c, h, w, gx.data_ptr(),
                                                        _stream_ptr(x.device)))
        return gx


def max_pool2(pool: torch.nn.MaxPool2d, x: torch.Tensor) -> torch.Tensor:
    """``pool(x)`` through the HIP kernels when it is the UNets' ``MaxPool2d(kernel_size=2, stride=2)`` on a float64 device
    tensor of at least 2 x 2 pixels; anything else goes to the torch module."""
    def two(v):
        return v in (2, (2, 2))
    if not (isinstance(pool, torch.nn.MaxPool2d) and two(pool.kernel_size) and two(pool.stride) and pool.padding in (0, (0, 0))
            and pool.dilation in (1, (1, 1)) and not pool.ceil_mode and not pool.return_indices and x.is_cuda
            and x.dtype == torch.float64 and x.dim() == 4 and x.numel() > 0 and x.shape[2] >= 2 and x.shape[3] >= 2):
        return pool(x)
    return _MaxPool2Function.apply(x)


def upsample2x(x: torch.Tensor, a_h: torch.Tensor, a_w: torch.Tensor) -> torch.Tensor:
    """``a_h x a_w^T`` per plane for float64 CUDA tensors (a_h: (2H, H), a_w: (2W, W), contiguous float64)."""
    return _Upsample2xFunction.apply(x, a_h, a_w)


def run_shift_sweep(circ: Circuit, inputs, angles: torch.Tensor, grad_out: torch.Tensor,
                    precision: str | None = None, with_inputs: bool = True,
                    max_dots_elems: int = 1 << 26):
    """Full parameter-shift sweep.  Returns (grad_angles (angles_shape, f64),
    grad_inputs (B, n) f64 or None).  One QNode round only."""
    precision = precision or _default_precision
    dtype = _DT[precision][1]
    device = angles.device
    x, ld, circ2 = _prep_inputs(circ, inputs, dtype, device)
    circ = circ2 or circ
    batch = x.shape[0] if x is not None else grad_out.shape[0]      # encoding "none": one row of grad_out per sample
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    table = prepare_gates(circ, angles, precision)
    g = grad_out.to(device=device, dtype=dtype).contiguous()
    want_inputs = with_inputs and circ.encoding in ("rz", "ry", "ry_blocks")
    total = lib.qiddm_num_shift_replicas(ctypes.byref(cs), 1 if want_inputs else 0)
    n_rot = lib.qiddm_num_rot_gates(ctypes.byref(cs))
    chunk = max(2, min(65534, (max_dots_elems // max(batch, 1)) // 2 * 2, total))
    if circ.n_qubits > 10:
        chunk = min(chunk, 256)   # bounds the tiled kernel's workspace (one slab per replica x block)
    w_sum = torch.empty(6 * n_rot, dtype=torch.float64, device=device)
    in_dots = []
    first = 0
    while first < total:
        cnt = min(chunk, total - first)
        dots = torch.empty(cnt, batch, dtype=dtype, device=device)
        ws_buf, ws_ptr, ws_bytes = _workspace(circ, precision, batch, cnt, device)
        _capi.check(lib.qiddm_forward_shifted(ctypes.byref(cs), x.data_ptr(), batch, ld,
                                              table.data_ptr(), g.data_ptr(), g.shape[1], first, cnt,
                                              dots.data_ptr(), ws_ptr, ws_bytes, _stream_ptr(device)))
        w_hi = min(first + cnt, 6 * n_rot)
        if first < w_hi:
            w_sum[first:w_hi] = dots[: w_hi - first].to(torch.float64).sum(dim=1)
        if first + cnt > 6 * n_rot:
            in_dots.append(dots[max(0, 6 * n_rot - first):].to(torch.float64))
        first += cnt
    pm = w_sum.view(n_rot, 3, 2)
    grad_angles = (0.5 * (pm[..., 0] - pm[..., 1])).view(circ.angles_shape)
    grad_inputs = None
    if want_inputs:
        d = torch.cat(in_dots, dim=0).view(circ.n_blocks, circ.n_qubits, 2, batch)
        grad_inputs = (0.5 * circ.enc_scale) * (d[:, :, 0] - d[:, :, 1]).sum(dim=0).transpose(0, 1)
    return grad_angles, grad_inputs


def run_adjoint(circ: Circuit, inputs, angles: torch.Tensor, grad_out: torch.Tensor,
                precision: str | None = None, with_inputs: bool = True):
    """Reverse-mode gradients of one QNode round (``qiddm_backward_adjoint``).
    Returns (grad_angles (angles_shape, f64), grad_inputs (B, n | n_features) f64 or None)."""
    precision = precision or _default_precision
    dtype = _DT[precision][1]
    device = angles.device
    if circ.n_rounds != 1:
        raise ValueError("run_adjoint differentiates one round")
    x, ld, circ2 = _prep_inputs(circ, inputs, dtype, device)
    circ = circ2 or circ
    batch = x.shape[0] if x is not None else grad_out.shape[0]      # encoding "none": one row of grad_out per sample
    lib = _capi.lib()
    cs = circ.c_struct(precision)
    table = prepare_gates(circ, angles, precision)
    g = grad_out.to(device=device, dtype=dtype).contiguous()
    n_rot = lib.qiddm_num_rot_gates(ctypes.byref(cs))
    n_part = lib.qiddm_adjoint_partials(ctypes.byref(cs), batch)
    if n_part < 0:
        _capi.check(-2)
    kp = torch.empty(n_part, n_rot, 8, dtype=dtype, device=device)
    gin = None
    gin_cols = circ.features if circ.encoding == "amplitude" else circ.n_qubits
    if with_inputs and circ.encoding != "none":
        gin = torch.empty(batch, gin_cols, dtype=dtype, device=device)
    x_ptr = 0 if x is None else x.data_ptr()
    if circ.n_qubits > 10:
        need = lib.qiddm_adjoint_workspace_bytes(ctypes.byref(cs), batch)
        ws = _scratch(_workspaces, "adjoint", need, device)
        _capi.check(lib.qiddm_backward_adjoint_wide(ctypes.byref(cs), x_ptr, batch, ld, table.data_ptr(),
                                                    g.data_ptr(), g.shape[1], kp.data_ptr(),
                                                    0 if gin is None else gin.data_ptr(), gin_cols, ws.data_ptr(),
                                                    ws.numel(), _stream_ptr(device)))
    else:
        _capi.check(lib.qiddm_backward_adjoint(ctypes.byref(cs), x_ptr, batch, ld, table.data_ptr(),
                                               g.data_ptr(), g.shape[1], kp.data_ptr(),
                                               0 if gin is None else gin.data_ptr(), gin_cols, _stream_ptr(device)))
    a64 = angles.detach().to(torch.float64).contiguous()
    ga = torch.empty(n_rot, 3, dtype=torch.float64, device=device)
    _capi.check(lib.qiddm_adjoint_finalize(ctypes.byref(cs), a64.data_ptr(), kp.data_ptr(), n_part,
                                           ga.data_ptr(), _stream_ptr(device)))
    return ga.reshape(circ.angles_shape), (None if gin is None else gin.to(torch.float64))


class _QNodeFunction(torch.autograd.Function):
    """One QNode round, differentiable by parameter shift (2 evaluations per
    gate angle, coefficient 1/2 -- the rule PennyLane applies for
    diff_method="parameter-shift", configured at reference nn/qdense.py:246)."""

    @staticmethod
    def forward(ctx, inputs, angles, circ, precision, diff_method):
        out = run_forward(circ, inputs, angles, precision)
        ctx.circ, ctx.precision, ctx.diff_method = circ, precision, diff_method
        ctx.save_for_backward(inputs if inputs is not None else torch.empty(0), angles)
        ctx.in_shape = None if inputs is None else tuple(inputs.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        inputs, angles = ctx.saved_tensors
        circ = ctx.circ
        need_in = ctx.needs_input_grad[0]
        use_adjoint = ctx.diff_method != "parameter-shift" and circ.n_qubits <= 16
        if need_in and circ.encoding == "amplitude" and not use_adjoint:
            raise NotImplementedError(
                "gradient w.r.t. amplitude-embedded features is not a gate parameter; "
                "parameter-shift cannot provide it (use diff_method='backprop')")
        if use_adjoint:
            ga, gi = run_adjoint(circ, inputs, angles, grad_out, ctx.precision, with_inputs=need_in)
        else:
            ga, gi = run_shift_sweep(circ, inputs, angles, grad_out, ctx.precision, with_inputs=need_in)
        grad_inputs = None
        if need_in and gi is not None:
            full = torch.zeros(inputs.shape if inputs.dim() == 2 else (1,) + tuple(inputs.shape),
                               dtype=inputs.dtype, device=inputs.device)
            full[:, : gi.shape[1]] = gi.to(inputs.dtype)
            grad_inputs = full.view(ctx.in_shape)
        return grad_inputs, ga.to(angles.dtype), None, None, None


def execute(circ: Circuit, inputs, angles: torch.Tensor, precision: str | None = None,
            diff_method: str = "backprop") -> torch.Tensor:
    """Differentiable execution of ``circ`` (all rounds).  With grad enabled the
    rounds run one QNode call at a time, as the reference chains them
    (nn/qdense.py:464-465); under ``torch.no_grad()`` they are fused in one launch.
    diff_method "parameter-shift": 2 kernel re-invocations per gate angle; anything else
    ("backprop", "adjoint", "best"): the adjoint kernels (register-resident for n <= 10, slab-resident for
    n = 11..16)."""
    precision = precision or _default_precision
    _require_device(angles, "the circuit weights")
    needs_grad = torch.is_grad_enabled() and (
        angles.requires_grad or (inputs is not None and inputs.requires_grad))
    if not needs_grad:
        return run_forward(circ, inputs, angles, precision)
    x = inputs
    out = None
    one = replace(circ, n_rounds=1)
    for r in range(circ.n_rounds):
        out = _QNodeFunction.apply(x, angles[r:r + 1], one, precision, diff_method)
        x = out
    return out
