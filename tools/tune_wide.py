#!/usr/bin/env python
"""Launch-shape sweep of the wide CZ forward (QIDDM_WIDE_WAVES x QIDDM_WIDE_GRID) at the C5 shard and at n = 12 / 14.
Run on the GPU box:  python tools/tune_wide.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd.circuit import Circuit, prepare_gates, run_forward  # noqa: E402

DEV = "cuda"


def t_us(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    torch.manual_seed(0)
    for n, B in ((16, 1024), (14, 1024), (12, 4096)):
        circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=2, n_blocks=6, sel_layers=2)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
        x = torch.rand(B, n, device=DEV)
        table = prepare_gates(circ, w, "f32")
        sweeps = 2 * (12 - 1)
        for waves in (2, 4, 8):
            for grid in (128, 256, 384, 512, 768, 1024):
                os.environ["QIDDM_WIDE_WAVES"] = str(waves)
                os.environ["QIDDM_WIDE_GRID"] = str(grid)
                us = t_us(lambda: run_forward(circ, x, w, "f32", table=table))
                gb = sweeps * 16 * (1 << n) * B / 1e9
                print(f"n={n} B={B} waves={waves} grid={grid:5d}: {us / 1e3:8.3f} ms  {gb / (us * 1e-6) / 1e3:6.2f} TB/s sweep traffic", flush=True)
    os.environ.pop("QIDDM_WIDE_WAVES", None)
    os.environ.pop("QIDDM_WIDE_GRID", None)


if __name__ == "__main__":
    main()
