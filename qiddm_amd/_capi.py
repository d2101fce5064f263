"""ctypes binding of the C ABI declared in ``include/qiddm_hip.h``.

The shared library is built in-tree by ``__graft_entry__.build()`` (or
``python -m qiddm_amd.build``) into ``qiddm_amd/lib/libqiddm_hip.so``.  There is
no CPU fallback: if the library is missing, ``lib()`` raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

# torch must come first: it ships its own HIP runtime (torch/lib/libamdhip64.so, soname
# libamdhip64.so.7).  Loading ours before torch's would bind the process to /opt/rocm's copy
# and the two runtimes then disagree about the device ("no ROCm-capable device").
import torch  # noqa: F401

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
# QIDDM_HIP_LIB: load another build of the same ABI (kernel experiments); default is the in-tree library
LIB_PATH = os.environ.get("QIDDM_HIP_LIB") or os.path.join(LIB_DIR, "libqiddm_hip.so")

QIDDM_OK = 0
ENC_NONE, ENC_AMPLITUDE, ENC_RZ, ENC_RY, ENC_RY_BLOCKS = 0, 1, 2, 3, 4
IMP_CNOT, IMP_CZ = 0, 1
MEAS_PROBS, MEAS_EXPZ = 0, 1
F32, F64 = 0, 1

# every symbol include/qiddm_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = (
    "qiddm_abi_version",
    "qiddm_max_qubits",
    "qiddm_last_error",
    "qiddm_set_stamp_buffer",
    "qiddm_num_rot_gates",
    "qiddm_gate_count",
    "qiddm_gate_table_elems",
    "qiddm_num_shift_replicas",
    "qiddm_workspace_bytes",
    "qiddm_prepare_gates",
    "qiddm_forward",
    "qiddm_forward_post",
    "qiddm_forward_shifted",
    "qiddm_adjoint_partials",
    "qiddm_backward_adjoint",
    "qiddm_adjoint_finalize",
    "qiddm_adjoint_workspace_bytes",
    "qiddm_backward_adjoint_wide",
    "qiddm_dense_forward",
    "qiddm_dense_sample",
    "qiddm_dense_sample_tables_bytes",
    "qiddm_dense_sample_prepare",
    "qiddm_dense_sample_lean_tables_bytes",
    "qiddm_dense_sample_lean_prepare",
    "qiddm_dense_sample_lean_check",
    "qiddm_dense_sample_lean",
    "qiddm_qconv_forward",
    "qiddm_qconv_backward",
    "qiddm_train_workspace_bytes",
    "qiddm_train_step",
    "qiddm_adam_step",
    "qiddm_circuit_unitary",
    "qiddm_circuit_unitary_wide",
    "qiddm_qconv_unitary_workspace_bytes",
    "qiddm_qconv_unitary_forward",
    "qiddm_amp_embed_rows",
    "qiddm_prob_post",
    "qiddm_maxpool2_forward",
    "qiddm_maxpool2_backward",
    "qiddm_conv1x1_forward",
    "qiddm_conv1x1_head_partials",
    "qiddm_conv1x1_head_backward",
    "qiddm_qconv_fold_features",
    "qiddm_qconv_train_rows",
    "qiddm_qconv_train_vectors",
    "qiddm_qconv_train_partials",
    "qiddm_qconv_train_backward",
    "qiddm_qconv_train_x32_ok",
    "qiddm_qconv_train_backward_x32",
    "qiddm_qconv_train_dx_elems",
    "qiddm_qconv_train_backward_dx",
    "qiddm_matrix_adjoint_partials",
    "qiddm_matrix_adjoint_workspace_bytes",
    "qiddm_matrix_adjoint",
    "qiddm_batchnorm_workspace_bytes",
    "qiddm_batchnorm_train_forward",
    "qiddm_batchnorm_backward",
    "qiddm_batchnorm_backward_stats",
    "qiddm_qconv_train_backward_bn",
    "qiddm_qconv_train_bn_ok",
    "qiddm_upsample2x_forward",
    "qiddm_upsample2x_backward",
    "qiddm_mixed_workspace_bytes",
    "qiddm_mixed_forward",
)


class CircuitStruct(ctypes.Structure):
    """``qiddm_circuit_t``."""

    _fields_ = [
        ("n_qubits", ctypes.c_int32),
        ("encoding", ctypes.c_int32),
        ("imprimitive", ctypes.c_int32),
        ("measure", ctypes.c_int32),
        ("n_rounds", ctypes.c_int32),
        ("n_blocks", ctypes.c_int32),
        ("sel_layers", ctypes.c_int32),
        ("n_features", ctypes.c_int32),
        ("dtype", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("enc_scale", ctypes.c_double),
        ("enc_offset", ctypes.c_double),
        ("pad_with", ctypes.c_double),
    ]


class QiddmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libqiddm_hip: {msg} (status {code})")
        self.code = code


_lock = threading.Lock()
_lib = None


def _declare(lib):
    P = ctypes.POINTER(CircuitStruct)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.qiddm_abi_version.restype = ctypes.c_int
    lib.qiddm_abi_version.argtypes = []
    lib.qiddm_max_qubits.restype = ctypes.c_int
    lib.qiddm_max_qubits.argtypes = []
    lib.qiddm_last_error.restype = ctypes.c_char_p
    lib.qiddm_last_error.argtypes = []
    lib.qiddm_set_stamp_buffer.restype = ctypes.c_int
    lib.qiddm_set_stamp_buffer.argtypes = [vp, i64]
    for name in ("qiddm_num_rot_gates", "qiddm_gate_count", "qiddm_gate_table_elems"):
        getattr(lib, name).restype = i64
        getattr(lib, name).argtypes = [P]
    lib.qiddm_num_shift_replicas.restype = i64
    lib.qiddm_num_shift_replicas.argtypes = [P, ctypes.c_int]
    lib.qiddm_prepare_gates.restype = ctypes.c_int
    lib.qiddm_prepare_gates.argtypes = [P, vp, vp, vp]
    lib.qiddm_forward.restype = ctypes.c_int
    lib.qiddm_forward.argtypes = [P, vp, i64, i64, vp, vp, i64, vp, i64, vp]
    lib.qiddm_forward_post.restype = ctypes.c_int
    lib.qiddm_forward_post.argtypes = [P, vp, i64, i64, vp, vp, i64, ctypes.c_int32, ctypes.c_double, vp]
    lib.qiddm_forward_shifted.restype = ctypes.c_int
    lib.qiddm_forward_shifted.argtypes = [P, vp, i64, i64, vp, vp, i64, i64, i64, vp, vp, i64, vp]
    lib.qiddm_workspace_bytes.restype = i64
    lib.qiddm_workspace_bytes.argtypes = [P, i64, i64]
    lib.qiddm_dense_forward.restype = ctypes.c_int
    lib.qiddm_dense_forward.argtypes = [P, vp, i64, i64, i64, vp, vp, vp, vp, vp, i64, ctypes.c_int32,
                                        ctypes.c_double, vp, i64, vp]
    lib.qiddm_dense_sample.restype = ctypes.c_int
    lib.qiddm_dense_sample.argtypes = [P, vp, i64, i64, i64, vp, vp, vp, vp, vp, i64, ctypes.c_int32,
                                       ctypes.c_double, ctypes.c_int32, vp, i64, i64, vp, vp]
    lib.qiddm_dense_sample_tables_bytes.restype = i64
    lib.qiddm_dense_sample_tables_bytes.argtypes = [P]
    lib.qiddm_dense_sample_prepare.restype = ctypes.c_int
    lib.qiddm_dense_sample_prepare.argtypes = [P, vp, vp, vp]
    lib.qiddm_dense_sample_lean_tables_bytes.restype = i64
    lib.qiddm_dense_sample_lean_tables_bytes.argtypes = [P]
    lib.qiddm_dense_sample_lean_prepare.restype = ctypes.c_int
    lib.qiddm_dense_sample_lean_prepare.argtypes = [P, vp, vp, vp, vp, vp, i64, vp, vp]
    lib.qiddm_dense_sample_lean_check.restype = ctypes.c_int
    lib.qiddm_dense_sample_lean_check.argtypes = [P, vp, vp]
    lib.qiddm_dense_sample_lean.restype = ctypes.c_int
    lib.qiddm_dense_sample_lean.argtypes = [P, vp, i64, i64, i64, vp, vp, vp, vp, ctypes.c_int32, ctypes.c_double,
                                            ctypes.c_int32, vp, i64, i64, vp, vp]
    lib.qiddm_adjoint_partials.restype = i64
    lib.qiddm_adjoint_partials.argtypes = [P, i64]
    lib.qiddm_backward_adjoint.restype = ctypes.c_int
    lib.qiddm_backward_adjoint.argtypes = [P, vp, i64, i64, vp, vp, i64, vp, vp, i64, vp]
    lib.qiddm_adjoint_workspace_bytes.restype = i64
    lib.qiddm_adjoint_workspace_bytes.argtypes = [P, i64]
    lib.qiddm_backward_adjoint_wide.restype = ctypes.c_int
    lib.qiddm_backward_adjoint_wide.argtypes = [P, vp, i64, i64, vp, vp, i64, vp, vp, i64, vp, i64, vp]
    lib.qiddm_adjoint_finalize.restype = ctypes.c_int
    lib.qiddm_adjoint_finalize.argtypes = [P, vp, vp, i64, vp, vp]
    lib.qiddm_qconv_forward.restype = ctypes.c_int
    lib.qiddm_qconv_forward.argtypes = [P, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, i64, vp, vp]
    lib.qiddm_qconv_backward.restype = ctypes.c_int
    lib.qiddm_qconv_backward.argtypes = [P, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, vp, i64, vp, vp, vp, vp]
    dbl = ctypes.c_double
    lib.qiddm_upsample2x_forward.restype = ctypes.c_int
    lib.qiddm_upsample2x_forward.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp]
    lib.qiddm_upsample2x_backward.restype = ctypes.c_int
    lib.qiddm_upsample2x_backward.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp]
    lib.qiddm_qconv_fold_features.restype = ctypes.c_int
    lib.qiddm_qconv_fold_features.argtypes = [vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, vp]
    lib.qiddm_qconv_train_rows.restype = ctypes.c_int
    lib.qiddm_qconv_train_rows.argtypes = [ctypes.c_int32, vp, ctypes.c_int32, i64, i64, ctypes.c_int32, vp, vp]
    lib.qiddm_qconv_train_vectors.restype = ctypes.c_int
    lib.qiddm_qconv_train_vectors.argtypes = [ctypes.c_int32, vp, i64, i64, i64, ctypes.c_int32, vp, vp, vp]
    lib.qiddm_qconv_train_partials.restype = ctypes.c_int64
    lib.qiddm_qconv_train_partials.argtypes = [i64, i64, i64, i64]
    lib.qiddm_qconv_train_backward.restype = ctypes.c_int
    lib.qiddm_qconv_train_backward.argtypes = [ctypes.c_int32, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, i64, vp,
                                               ctypes.c_int32, vp, vp, vp, vp]
    lib.qiddm_qconv_train_x32_ok.restype = ctypes.c_int32
    lib.qiddm_qconv_train_x32_ok.argtypes = [i64, i64, i64, i64, i64, i64, i64, i64, i64, ctypes.c_int32]
    lib.qiddm_qconv_train_backward_x32.restype = ctypes.c_int
    lib.qiddm_qconv_train_backward_x32.argtypes = lib.qiddm_qconv_train_backward.argtypes
    lib.qiddm_qconv_train_dx_elems.restype = ctypes.c_int64
    lib.qiddm_qconv_train_dx_elems.argtypes = [ctypes.c_int32, i64, i64, i64, i64, i64, i64, i64, i64, i64, ctypes.c_int32]
    lib.qiddm_qconv_train_backward_dx.restype = ctypes.c_int
    lib.qiddm_qconv_train_backward_dx.argtypes = [ctypes.c_int32, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, i64, i64,
                                                  vp, ctypes.c_int32, vp, vp, vp, vp]
    lib.qiddm_matrix_adjoint_partials.restype = ctypes.c_int64
    lib.qiddm_matrix_adjoint_partials.argtypes = [i64]
    lib.qiddm_matrix_adjoint_workspace_bytes.restype = ctypes.c_int64
    lib.qiddm_matrix_adjoint_workspace_bytes.argtypes = [P, i64]
    lib.qiddm_matrix_adjoint.restype = ctypes.c_int
    lib.qiddm_matrix_adjoint.argtypes = [P, vp, vp, i64, vp, vp, vp, i64, vp]
    lib.qiddm_batchnorm_workspace_bytes.restype = ctypes.c_int64
    lib.qiddm_batchnorm_workspace_bytes.argtypes = [i64, i64, i64]
    lib.qiddm_batchnorm_train_forward.restype = ctypes.c_int
    lib.qiddm_batchnorm_train_forward.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, dbl, dbl, vp, vp, vp, vp, i64, vp]
    lib.qiddm_batchnorm_backward_stats.restype = ctypes.c_int
    lib.qiddm_batchnorm_backward_stats.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.qiddm_qconv_train_bn_ok.restype = ctypes.c_int32
    lib.qiddm_qconv_train_bn_ok.argtypes = [i64, i64, i64, i64, i64, i64, i64, i64, i64, ctypes.c_int32]
    lib.qiddm_qconv_train_backward_bn.restype = ctypes.c_int
    lib.qiddm_qconv_train_backward_bn.argtypes = [ctypes.c_int32, vp, i64, i64, i64, i64, i64, i64, i64, i64, vp, vp, vp,
                                                  i64, vp, ctypes.c_int32, vp, vp, vp, vp, vp]
    lib.qiddm_batchnorm_backward.restype = ctypes.c_int
    lib.qiddm_batchnorm_backward.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.qiddm_train_workspace_bytes.restype = ctypes.c_int64
    lib.qiddm_train_workspace_bytes.argtypes = [P, i64, ctypes.c_int32, ctypes.c_int32]
    lib.qiddm_train_step.restype = ctypes.c_int
    lib.qiddm_train_step.argtypes = [P, ctypes.POINTER(TrainArgs), vp, i64, vp]
    lib.qiddm_circuit_unitary.restype = ctypes.c_int
    lib.qiddm_circuit_unitary.argtypes = [P, vp, vp, vp]
    lib.qiddm_circuit_unitary_wide.restype = ctypes.c_int
    lib.qiddm_circuit_unitary_wide.argtypes = [P, vp, vp, vp]
    lib.qiddm_qconv_unitary_workspace_bytes.restype = i64
    lib.qiddm_qconv_unitary_workspace_bytes.argtypes = [ctypes.c_int32, i64, i64, i64, i64]
    lib.qiddm_qconv_unitary_forward.restype = ctypes.c_int
    lib.qiddm_qconv_unitary_forward.argtypes = [ctypes.c_int32, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64,
                                                i64, ctypes.c_int32, ctypes.POINTER(BatchNormStruct), ctypes.c_int32, vp, vp,
                                                i64, vp]
    lib.qiddm_mixed_workspace_bytes.restype = i64
    lib.qiddm_mixed_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, i64, ctypes.c_int32]
    lib.qiddm_mixed_forward.restype = ctypes.c_int
    lib.qiddm_mixed_forward.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(MixedOp), ctypes.c_int32, vp, i64,
                                        ctypes.c_int32, vp, i64, ctypes.c_int32, ctypes.c_double, ctypes.c_double, vp,
                                        ctypes.c_int32, ctypes.c_int32, i64, vp, i64, vp, i64, vp]
    lib.qiddm_amp_embed_rows.restype = ctypes.c_int
    lib.qiddm_amp_embed_rows.argtypes = [vp, i64, i64, i64, ctypes.c_int32, ctypes.c_double, ctypes.c_double, vp, vp]
    lib.qiddm_prob_post.restype = ctypes.c_int
    lib.qiddm_prob_post.argtypes = [vp, i64, i64, ctypes.c_double, vp, vp]
    lib.qiddm_maxpool2_forward.restype = ctypes.c_int
    lib.qiddm_maxpool2_forward.argtypes = [vp, i64, i64, i64, vp, vp]
    lib.qiddm_maxpool2_backward.restype = ctypes.c_int
    lib.qiddm_maxpool2_backward.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    lib.qiddm_conv1x1_forward.restype = ctypes.c_int
    lib.qiddm_conv1x1_forward.argtypes = [vp, vp, vp, i64, i64, i64, i64, vp, vp]
    lib.qiddm_conv1x1_head_partials.restype = ctypes.c_int64
    lib.qiddm_conv1x1_head_partials.argtypes = [i64, i64]
    lib.qiddm_conv1x1_head_backward.restype = ctypes.c_int
    lib.qiddm_conv1x1_head_backward.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, vp, vp, vp]
    lib.qiddm_adam_step.restype = ctypes.c_int
    lib.qiddm_adam_step.argtypes = [ctypes.POINTER(AdamTensor), ctypes.c_int32, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_double, vp, vp]


class TrainArgs(ctypes.Structure):
    """``qiddm_train_args_t``."""

    _fields_ = [
        ("x", ctypes.c_void_p), ("noise", ctypes.c_void_p), ("schedule", ctypes.c_void_p),
        ("x_ld", ctypes.c_int64), ("noise_ld", ctypes.c_int64), ("batch", ctypes.c_int64),
        ("pixels", ctypes.c_int32), ("tau", ctypes.c_int32), ("goal", ctypes.c_int32),
        ("train_quantum", ctypes.c_int32),
        ("w_down", ctypes.c_void_p), ("b_down", ctypes.c_void_p), ("angles", ctypes.c_void_p),
        ("w_up", ctypes.c_void_p), ("b_up", ctypes.c_void_p),
        ("loss", ctypes.c_void_p),
        ("g_w_down", ctypes.c_void_p), ("g_b_down", ctypes.c_void_p), ("g_angles", ctypes.c_void_p),
        ("g_w_up", ctypes.c_void_p), ("g_b_up", ctypes.c_void_p),
        ("recon", ctypes.c_void_p), ("elem_loss", ctypes.c_void_p), ("rng_state", ctypes.c_void_p),
    ]


MIX_ZERO, MIX_AMP_EMBED, MIX_PHASE, MIX_RY, MIX_GATE, MIX_CZ, MIX_CNOT, MIX_PHASE_DAMP, MIX_AMP_DAMP, MIX_DEPOL = range(10)


class MixedOp(ctypes.Structure):
    """``qiddm_mixed_op_t``."""

    _fields_ = [("kind", ctypes.c_int32), ("wire", ctypes.c_int32), ("a", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("p", ctypes.c_double), ("scale", ctypes.c_double)]


class BatchNormStruct(ctypes.Structure):
    """``qiddm_batchnorm_t``."""

    _fields_ = [("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("running_mean", ctypes.c_void_p),
                ("running_var", ctypes.c_void_p), ("eps", ctypes.c_double)]


class AdamTensor(ctypes.Structure):
    """``qiddm_adam_tensor_t``."""

    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("step", ctypes.c_void_p), ("numel", ctypes.c_int64), ("dtype", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


def _preload_torch_hip_runtime():
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Load (once) and return the C-ABI library.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: the HIP extension has not been built. "
                    "Run `python -c 'import __graft_entry__ as g; g.build()'` (or "
                    "`python -m qiddm_amd.build`) from the repo root. "
                    "qiddm_amd has no CPU fallback for the quantum layers."
                )
            _preload_torch_hip_runtime()
            handle = ctypes.CDLL(LIB_PATH)
            _declare(handle)
            if handle.qiddm_abi_version() != 1:
                raise RuntimeError("libqiddm_hip.so ABI version mismatch; rebuild it")
            _lib = handle
    return _lib


def check(status: int):
    if status != QIDDM_OK:
        raise QiddmError(status, lib().qiddm_last_error().decode("utf-8", "replace"))
