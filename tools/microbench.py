#!/usr/bin/env python
"""Kernel-level timings (HIP events around hipGraph-replayed back-to-back launches).
Run on the GPU box:  python tools/microbench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd.circuit import Circuit, dense_forward, prepare_gates, run_forward  # noqa: E402

DEV = "cuda"


def timeit(fn, launches=100, reps=5):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * launches)


def main():
    torch.manual_seed(0)
    rows = []
    for n, N, L, S, P in [(8, 1, 1, 14, 784), (8, 1, 1, 1, 784), (8, 2, 6, 2, 784), (10, 2, 9, 2, 784), (4, 1, 1, 2, 64)]:
        for B in (256, 4096, 65536):
            for prec in ("f32", "f64"):
                if prec == "f64" and B > 4096:
                    continue
                circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=N,
                               n_blocks=L, sel_layers=S)
                w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
                x = torch.randn(B, n, device=DEV)
                table = prepare_gates(circ, w, prec)
                t_c = timeit(lambda: run_forward(circ, x, w, prec, table=table))
                img = torch.rand(B, P, dtype=torch.float64, device=DEV)
                wd = torch.randn(n, P, dtype=torch.float64, device=DEV) * 0.05
                bd = torch.randn(n, dtype=torch.float64, device=DEV)
                wu = torch.randn(P, n, dtype=torch.float64, device=DEV)
                bu = torch.randn(P, dtype=torch.float64, device=DEV)
                t_d = timeit(lambda: dense_forward(circ, img, wd, bd, w, wu, bu, prec))
                g = circ.gate_count()
                rows.append((n, N, L, S, B, prec, g, t_c, t_d, B * g / t_c * 1e-3, B / t_d))
                print(f"n={n} N={N} L={L} S={S} B={B:6d} {prec}: G={g:4d} circuit {t_c:9.2f} us  dense {t_d:9.2f} us "
                      f"| {B * g / t_c * 1e-3:8.2f} G gate-apps/s (circuit) | {B / t_d:8.2f} M img/s (dense)",
                      flush=True)


if __name__ == "__main__":
    main()
