"""A/B: circuit_kernel (all paths) against circuit_folded_kernel (tangent form) at small batches.
QIDDM_LEAN_ABOVE=0 python tools/ab_lean_threshold.py   vs   python tools/ab_lean_threshold.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd.circuit import Circuit, prepare_gates, run_forward  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from microbench import timeit  # noqa: E402
torch.manual_seed(0)
print("QIDDM_LEAN_ABOVE =", os.environ.get("QIDDM_LEAN_ABOVE", "1024 (default)"))
for n, N, L, S, meas in [(8, 1, 1, 14, "expz"), (8, 2, 6, 2, "expz"), (9, 2, 6, 2, "expz"), (10, 2, 9, 2, "probs")]:
    for B in (64, 256, 512, 1024, 2048):
        circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_rounds=N, n_blocks=L, sel_layers=S)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).cuda()
        x = torch.randn(B, n, device="cuda")
        table = prepare_gates(circ, w, "f32")
        t = timeit(lambda: run_forward(circ, x, w, "f32", table=table), launches=50, reps=3)
        print(f"n={n} N={N} L={L} S={S} {meas} B={B:5d}: {t:8.2f} us", flush=True)
