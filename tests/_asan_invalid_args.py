"""Run INSIDE a python that has the AddressSanitizer runtime preloaded and QIDDM_HIP_LIB pointing at the ASan build of
the C ABI's host side (tests/test_capi_asan.py starts it).  Drives the argument validation of the entry points with
invalid descriptors, NULL pointers and inconsistent sizes: every call must come back with a negative status and a
reason -- and AddressSanitizer must have nothing to say (it aborts the process otherwise).  No GPU is touched: every
case is rejected before the first HIP call."""
import ctypes
import sys

from qiddm_amd import _capi
from qiddm_amd.circuit import Circuit

lib = _capi.lib()
P = ctypes.byref
n_checked = 0


def rejected(status, needle=None):
    global n_checked
    assert status < 0, status
    msg = lib.qiddm_last_error()
    assert msg, "no reason recorded"
    if needle is not None:
        assert needle in msg, (needle, msg)
    n_checked += 1


# ---- descriptors (the cases of tests/test_capi_symbols.py::test_invalid_descriptors_are_rejected_with_a_reason) ----
cs = Circuit(17, "rz").c_struct("f32")
rejected(lib.qiddm_gate_count(P(cs)), b"exceeds")
cs = Circuit(2, "amplitude", "CNOT", "probs", n_features=5).c_struct("f32")
rejected(lib.qiddm_gate_count(P(cs)), b"Features must be of length 4 or smaller")
cs = Circuit(3).c_struct("f32")
cs.sel_layers = 0
rejected(lib.qiddm_num_rot_gates(P(cs)))
for field, value in (("n_qubits", 0), ("n_qubits", -3), ("encoding", 9), ("imprimitive", 7), ("measure", -1),
                     ("n_rounds", 0), ("n_blocks", -1), ("dtype", 5)):
    cs = Circuit(4, "rz", "CZ", "expz").c_struct("f32")
    setattr(cs, field, value)
    rejected(lib.qiddm_gate_count(P(cs)))
    rejected(lib.qiddm_gate_table_elems(P(cs)))
    rejected(lib.qiddm_prepare_gates(P(cs), None, None, None))
    rejected(lib.qiddm_forward(P(cs), None, 4, 4, None, None, 4, None, 0, None))
    rejected(lib.qiddm_dense_sample_tables_bytes(P(cs)))

# ---- NULL pointers / sizes on valid descriptors ------------------------------------------------------------------
ok = Circuit(8, "rz", "CZ", "expz", 1, 1, 14).c_struct("f32")
rejected(lib.qiddm_prepare_gates(P(ok), None, None, None))
rejected(lib.qiddm_forward(P(ok), None, 16, 8, None, None, 8, None, 0, None))
rejected(lib.qiddm_forward(P(ok), None, -1, 8, None, None, 8, None, 0, None))
rejected(lib.qiddm_forward_shifted(P(ok), None, 16, 8, None, None, 8, 0, 4, None, None, 0, None))
rejected(lib.qiddm_dense_sample(P(ok), None, 4, 784, 784, None, None, None, None, None, 784, 0, 1.0, 2, None, 784,
                                4 * 784, None, None))
rejected(lib.qiddm_dense_sample(P(ok), None, -4, 784, 784, None, None, None, None, None, 784, 0, 1.0, 2, None, 784,
                                4 * 784, None, None))
rejected(lib.qiddm_dense_sample(P(ok), None, 4, 784, 784, None, None, None, None, None, 100, 1, 1.0, 2, None, 784,
                                4 * 784, None, None), b"out_features == in_features")
rejected(lib.qiddm_dense_sample(P(ok), None, 4, 784, 784, None, None, None, None, None, 784, 3, 1.0, 2, None, 784,
                                4 * 784, None, None), b"post_mode")
rejected(lib.qiddm_dense_sample_prepare(P(ok), None, None, None))
wide = Circuit(16, "rz", "CZ", "expz", 2, 6, 2).c_struct("f32")
rejected(lib.qiddm_dense_sample_tables_bytes(P(wide)))
# a 16-qubit forward without its workspace
rejected(lib.qiddm_forward(P(wide), None, 8, 16, None, None, 16, None, 0, None))
cnot = Circuit(10, "amplitude", "CNOT", "probs", 1, 1, 60, n_features=784, pad_with=0.1).c_struct("f32")
rejected(lib.qiddm_dense_sample_tables_bytes(P(cnot)))

print(f"[asan] {n_checked} invalid calls rejected cleanly")
sys.exit(0)
