#!/usr/bin/env python
"""Diagnostic: where one wave of qiddm_dense_forward spends its cycles (s_memtime stamps of
block 0 / thread 0).  Not a timing of the product path -- the stamps serialise it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
from qiddm_amd import _capi  # noqa: E402
_capi.check(_capi.lib().qiddm_set_stamp_buffer(buf.data_ptr(), buf.numel()))
from qiddm_amd.circuit import Circuit, dense_forward  # noqa: E402
for (n, N, L, S, P, B) in [(8, 1, 1, 14, 784, 256), (8, 1, 1, 1, 784, 256), (8, 1, 1, 14, 784, 4096)]:
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=N, n_blocks=L, sel_layers=S)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).cuda()
    img = torch.rand(B, P, dtype=torch.float64, device="cuda")
    wd = torch.randn(n, P, dtype=torch.float64, device="cuda") * 0.05
    bd = torch.randn(n, dtype=torch.float64, device="cuda")
    wu = torch.randn(P, n, dtype=torch.float64, device="cuda")
    bu = torch.randn(P, dtype=torch.float64, device="cuda")
    for _ in range(3):
        dense_forward(circ, img, wd, bd, w, wu, bu, "f32")
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    d = [t[i + 1] - t[i] for i in range(4)]
    print(f"n={n} S={S} B={B}: stage {d[0]}  linear_down {d[1]}  circuit {d[2]}  linear_up {d[3]}  (cycles @100MHz? raw s_memtime ticks)")
