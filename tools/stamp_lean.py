#!/usr/bin/env python
"""Diagnostic: where workgroup 0 of the lean 8-qubit sampler (dense_quad8_kernel) spends the cycles of a step
(s_memtime stamps of thread 0 in the launch's second step).  Not a timing of the product path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiddm_amd  # noqa: E402
from qiddm_amd import _capi, models, nn, noise  # noqa: E402

buf = torch.zeros(8, dtype=torch.int64, device="cuda")
_capi.check(_capi.lib().qiddm_set_stamp_buffer(buf.data_ptr(), buf.numel()))
x = (torch.rand(256, 1, 28, 28, dtype=torch.double) * 0.75 + 0.5).cuda()
for prec in ("f32", "f64"):
    qiddm_amd.set_default_precision(prec)
    for name, ctor, layers in (("QNN_noise(784,8,14)", lambda: nn.QNN_noise(784, 8, 14), 13),
                               ("QIDDM_LL_noise(784,8,6,2)", lambda: nn.QIDDM_LL_noise(784, 8, 6, 2), 11)):
        torch.manual_seed(42)
        net = ctor().to("cuda", dtype=torch.double).eval()
        diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (28, 28)).to("cuda", dtype=torch.double).eval()
        with torch.no_grad():
            for _ in range(3):
                diff.denoise_steps(x, 4)
        torch.cuda.synchronize()
        t = buf.cpu().tolist()
        print(f"{prec} {name}: setup {t[1]-t[0]}  angles {t[3]-t[2]}  round prologue {t[7]-t[3]}  "
              f"{layers} layers of round 0 {t[4]-t[7]} ({(t[4]-t[7])/layers:.0f} each)  rest of rounds + read-out {t[5]-t[4]}  "
              f"linear_up {t[6]-t[5]}  step {t[6]-t[2]} ticks")
_capi.lib().qiddm_set_stamp_buffer(None, 0)
