#!/usr/bin/env python
"""Timings for the other BASELINE configurations (C1, C3, C4, C5 shapes) and for the backward passes.
Run on the GPU box:  python tools/microbench_wide.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd.circuit import Circuit, prepare_gates, run_adjoint, run_forward, run_shift_sweep  # noqa: E402

DEV = "cuda"


def t_eager(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    torch.manual_seed(0)
    print("== forward, f32 ==")
    for tag, n, enc, imp, meas, N, L, S, feat, B in [
        ("C1 QNN_noise(64,4,2)", 4, "rz", "CZ", "expz", 1, 1, 2, None, 32),
        ("C3 differN_noise(28,9,2)", 10, "rz", "CZ", "probs", 2, 9, 2, None, 1024),
        ("C3 QDenseUndirected_old_noise(60,28)", 10, "amplitude", "CNOT", "probs", 1, 1, 60, 784, 1024),
        ("C4 QConv2d 12q (C_in=256,k=3,qdepth=3), 512 images x 1024 px / 64", 12, "amplitude", "CNOT", "probs", 1, 1, 3, 2304, 8192),
        ("C5 LL-style (2352,16,6,2), 1024 per GPU", 16, "rz", "CZ", "expz", 2, 6, 2, None, 1024),
        ("n=12 LL-style", 12, "rz", "CZ", "expz", 2, 6, 2, None, 4096),
        ("n=14 LL-style", 14, "rz", "CZ", "expz", 2, 6, 2, None, 1024),
    ]:
        circ = Circuit(n_qubits=n, encoding=enc, imprimitive=imp, measure=meas, n_rounds=N, n_blocks=L, sel_layers=S,
                       n_features=feat or 0, pad_with=0.1)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
        x = torch.rand(B, feat or n, device=DEV)
        table = prepare_gates(circ, w, "f32")
        t = t_eager(lambda: run_forward(circ, x, w, "f32", table=table), iters=5 if n > 12 else 20)
        g = circ.gate_count()
        eq = B * circ.algorithmic_bytes_per_sample("f32") / t / 1e12
        print(f"{tag:70s} B={B:6d} G={g:5d}: {t * 1e3:9.3f} ms  {B / t:12.0f} circuits/s  {B * g / t / 1e9:7.2f} G gate-apps/s "
              f" {eq:8.2f} TB/s HBM-equivalent", flush=True)
    print("== backward (one round), f32 ==")
    for tag, n, L, S, B in [("QNN_noise(784,8,14) round", 8, 1, 14, 2560), ("LL(784,8,6,2) round", 8, 6, 2, 2560),
                            ("differN(28,9,2) round", 10, 9, 2, 1024), ("12q LL round (wide adjoint)", 12, 6, 2, 1024),
                            ("C5 16q LL round (wide adjoint)", 16, 6, 2, 128), ("C5 16q LL round, per-GPU shard", 16, 6, 2, 1024),
                            ("14q LL round", 14, 6, 2, 1024)]:
        meas = "probs" if n == 10 else "expz"
        circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_blocks=L, sel_layers=S)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
        x = torch.rand(B, n, device=DEV)
        gout = torch.randn(B, circ.out_cols, device=DEV)
        tf = t_eager(lambda: run_forward(circ, x, w, "f32"))
        ta = t_eager(lambda: run_adjoint(circ, x, w, gout, "f32"))
        if B > 256 and n > 12:      # the parameter-shift sweep of a 16-qubit shard takes minutes: skip
            print(f"{tag:30s} B={B}: forward {tf * 1e3:8.3f} ms  adjoint {ta * 1e3:8.3f} ms", flush=True)
            continue
        ts = t_eager(lambda: run_shift_sweep(circ, x, w, gout, "f32"), iters=1 if n > 10 else 2, warm=0 if n > 10 else 1)
        print(f"{tag:30s} B={B}: forward {tf * 1e3:8.3f} ms  adjoint {ta * 1e3:8.3f} ms  parameter-shift {ts * 1e3:9.2f} ms "
              f"({ts / ta:6.1f}x)", flush=True)


if __name__ == "__main__":
    main()
