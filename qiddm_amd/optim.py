"""Adam with every parameter tensor updated in one launch (``qiddm_adam_step``).

Same update rule and defaults as ``torch.optim.Adam`` -- what the reference harness constructs
(src/mnist_exm.py:170, ``optim.Adam(diff.parameters(), lr=lr)``) -- without ``amsgrad`` / ``maximize``.  The step
counter lives on the device, so the optimizer can be recorded into a HIP graph together with the fused training
step (``qiddm_amd.trainer.GraphedTrainStep``): one node instead of the ~10 a foreach implementation needs.
State keys (``step``, ``exp_avg``, ``exp_avg_sq``) match torch's (``step`` is an int64 device scalar).  There is no CPU path: parameters must live on the GPU.
"""
from __future__ import annotations

import ctypes

import torch

from . import _capi


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        # `capturable` is informational: the update never touches the host
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True))

    def _sync(self, device):
        table = self.__dict__.setdefault("_sync_words", {})
        t = table.get(device)
        if t is None:
            t = table[device] = torch.zeros((), dtype=torch.int32, device=device)
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _capi.lib()
        for group in self.param_groups:
            entries, device = [], None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam updates parameters on the GPU only (no CPU path)")
                if p.dtype not in (torch.float32, torch.float64):
                    raise TypeError(f"FusedAdam supports float32/float64 parameters, got {p.dtype}")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous parameters")
                device = device or p.device
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.int64, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad
                if g.dtype != p.dtype or not g.is_contiguous():
                    g = g.to(p.dtype).contiguous()
                entries.append((p, g, st["exp_avg"], st["exp_avg_sq"], st["step"]))
            if not entries:
                continue
            arr = (_capi.AdamTensor * len(entries))()
            for a, (p, g, m, v, n) in zip(arr, entries):
                a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                a.step = n.data_ptr()
                a.numel = p.numel()
                a.dtype = _capi.F64 if p.dtype == torch.float64 else _capi.F32
            b1, b2 = group["betas"]
            stream = torch.cuda.current_stream(device).cuda_stream
            _capi.check(lib.qiddm_adam_step(arr, len(entries), float(group["lr"]), float(b1), float(b2),
                                            float(group["eps"]), float(group["weight_decay"]),
                                            self._sync(device).data_ptr(), ctypes.c_void_p(stream)))
            self._keepalive = entries      # until the stream has consumed them
            for p, *_ in entries:          # the launch wrote the parameters behind autograd's back
                torch.autograd.graph.increment_version(p)
        return loss

    def reset_state(self):
        """Zero the moments and step counters in place (addresses recorded in a graph stay valid)."""
        for st in self.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
