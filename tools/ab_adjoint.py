"""Time the adjoint backward (qiddm_backward_adjoint + finalize) of one round; run with and without QIDDM_NO_FOLD=1."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd.circuit import Circuit, run_adjoint

torch.manual_seed(0)
for n, L, S, B in ((8, 1, 14, 2560), (8, 6, 2, 2560), (9, 9, 2, 1024), (10, 9, 2, 1024), (6, 14, 2, 2560), (4, 1, 2, 4096)):
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=1, n_blocks=L, sel_layers=S)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).cuda()
    x = torch.randn(B, n, dtype=torch.float64).cuda()
    g = torch.randn(B, n, dtype=torch.float64).cuda()
    for _ in range(3):
        run_adjoint(circ, x, w, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run_adjoint(circ, x, w, g)
    torch.cuda.synchronize()
    print(f"n={n} L={L} S={S} B={B}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per backward "
          f"({'general' if os.environ.get('QIDDM_NO_FOLD') else 'folded'})")
