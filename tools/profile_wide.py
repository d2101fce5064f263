#!/usr/bin/env python
"""Launches of the wide / backward kernels the headline bench does not exercise, for rocprofv3
(`--kernel-trace --stats`, then separate `--pmc` passes).  Run on the GPU box:

    rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/profile_wide.py [what ...]

what: c5 (n=16 forward, 1024 samples: wide_cz_kernel; QIDDM_WIDE_TILED=1 -> the generic tiled kernel)  c4gemm (12-qubit QConv2d eval GEMM)  adjoint (n=8 / n=10 backward)
      wideadj (n=12 / n=16 reverse sweep, 1024 samples, one round: wide_cz_adjoint_kernel)  engine (circuit_kernel<float,8> / <float,10> at batch 65536, dense_quad at 256)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiddm_amd.circuit import Circuit, prepare_gates, run_adjoint, run_forward  # noqa: E402

DEV = "cuda"


def c5(iters=3):
    torch.manual_seed(3)
    circ = Circuit(n_qubits=16, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=2, n_blocks=6, sel_layers=2)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
    x = torch.randn(1024, 16, device=DEV)
    table = prepare_gates(circ, w, "f32")
    for _ in range(iters):
        run_forward(circ, x, w, "f32", table=table)


def c4gemm(iters=3):
    from qiddm_amd import nn
    torch.manual_seed(42)
    conv = nn.QConv2d(256, 256, qdepth=3).to(DEV).eval()
    xc = torch.rand(64, 256, 32, 32, dtype=torch.double, device=DEV)
    with torch.no_grad():
        for _ in range(iters):
            conv(xc)


def adjoint(iters=3):
    for n, L, S, B, meas in ((8, 1, 14, 2560, "expz"), (8, 6, 2, 2560, "expz"), (10, 9, 2, 1024, "probs")):
        circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_blocks=L, sel_layers=S)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
        x = torch.rand(B, n, device=DEV)
        g = torch.randn(B, circ.out_cols, device=DEV)
        for _ in range(iters):
            run_adjoint(circ, x, w, g, "f32")


def wideadj(iters=2):
    for n, B in ((12, 1024), (16, 1024)):
        circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_blocks=6, sel_layers=2)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
        x = torch.rand(B, n, device=DEV)
        g = torch.randn(B, circ.out_cols, device=DEV)
        for _ in range(iters):
            run_adjoint(circ, x, w, g, "f32")


def engine(iters=3):
    for n, N, L, S, meas, B in ((8, 1, 1, 14, "expz", 65536), (8, 2, 6, 2, "expz", 65536), (10, 2, 9, 2, "probs", 65536)):
        circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure=meas, n_rounds=N, n_blocks=L, sel_layers=S)
        w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(DEV)
        x = torch.rand(B, n, device=DEV)
        table = prepare_gates(circ, w, "f32")
        for _ in range(iters):
            run_forward(circ, x, w, "f32", table=table)
    from qiddm_amd import models, nn, noise
    torch.manual_seed(42)
    net = nn.QNN_noise(784, 8, 14)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (28, 28)).to(DEV, dtype=torch.double).eval()
    xi = (torch.rand(256, 1, 28, 28, dtype=torch.double) * 0.75 + 0.5).to(DEV)
    with torch.no_grad():
        for _ in range(iters * 3):
            diff.denoise_steps(xi, 15)


if __name__ == "__main__":
    what = sys.argv[1:] or ["c5", "c4gemm", "adjoint", "wideadj", "engine"]
    for name in what:
        globals()[name]()
    torch.cuda.synchronize()
    print("profiled:", " ".join(what))
