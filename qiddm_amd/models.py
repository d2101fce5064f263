"""The denoise loop around the quantum layers (reference src/models.py:8-150)."""
from __future__ import annotations

import typing

import torch

try:  # progress bars are optional glue
    import tqdm
except Exception:  # pragma: no cover
    tqdm = None


class Diffusion(torch.nn.Module):
    """``Diffusion(net, noise_f, prediction_goal, shape, loss)``.

    * ``train()`` mode: ``diff(x=x, T=tau, verbose=bool)`` runs one training step **including
      ``.backward()``** (reference src/models.py:67, 99) and returns ``(loss, recon)`` or
      ``(loss,)``.
    * ``eval()`` mode: ``diff(x=first_x, n_iters=...)`` == ``diff.sample(...)``.
    * ``state_dict()`` keys are prefixed ``net.`` (checkpoint contract).
    """

    def __init__(self, net: torch.nn.Module, noise_f, prediction_goal: str,
                 shape: typing.Tuple[int, int],
                 loss: torch.nn.Module = torch.nn.MSELoss(reduction="none")) -> None:
        super().__init__()
        self.net = net
        self.prediction_goal = prediction_goal
        self.add_noise = noise_f
        self.width, self.height = shape
        self.loss = loss

    def forward(self, x=None, **kwargs):
        if not self.training:
            return self.sample(first_x=x, **kwargs)
        if self.prediction_goal == "data":
            return self.run_training_step_data(x, **kwargs)
        return self.run_training_step_noise(x, **kwargs)

    # -- training ---------------------------------------------------------------
    def _noisy_clean_pairs(self, x, T):
        """(batch*T, 1, W, H) noisy inputs and their one-step-cleaner targets
        (reference src/models.py:45-63)."""
        whole = self.add_noise(x, tau=T + 1, decay_mod=3.0).reshape(x.shape[0], T + 1, -1)
        noisy = whole[:, 1:, :].reshape(-1, 1, self.width, self.height)
        clean = whole[:, :-1, :].reshape(-1, 1, self.width, self.height)
        return noisy, clean

    def _fused_training_step(self, x, T, verbose):
        """The whole step on the device in three launches when the net offers it (``fused_train_step``) and the
        noising schedule / loss are the reference's own (``add_normal_noise_multiple``, ``MSELoss``); None
        otherwise.  Same return values as the eager methods below."""
        fused = getattr(self.net, "fused_train_step", None)
        field_f = getattr(self.add_noise, "noise_field", None)
        sched_f = getattr(self.add_noise, "schedule", None)
        if fused is None or field_f is None or sched_f is None or not torch.is_tensor(x) or not x.is_cuda \
                or x.dim() != 2 or x.shape[1] != self.width * self.height \
                or type(self.loss) is not torch.nn.MSELoss or self.loss.reduction not in ("mean", "none"):
            return None
        elementwise = self.loss.reduction == "none"
        schedule = sched_f(T + 1, 3.0, x.device).reshape(-1)
        rng = getattr(self.add_noise, "rng_state", None)     # set: the field is generated inside the launch
        res = fused(x, field_f(x), schedule, self.prediction_goal, want_recon=verbose,
                    want_elem_loss=verbose and elementwise, **({} if rng is None else {"rng_state": rng}))
        if res is None:
            return None
        loss = res["loss"]
        if not verbose:
            return (loss,)          # a mean of squares: the reference's .abs() (:71) is the identity on it
        shape = (-1, 1, self.width, self.height)
        recon = res["recon"].reshape(shape)
        batch_loss = res["elem_loss"].reshape(shape) if elementwise else loss
        return (batch_loss.abs(), recon.abs()) if self.prediction_goal == "data" else (batch_loss, recon)

    def run_training_step_data(self, x: torch.Tensor, **kwargs):
        out = self._fused_training_step(x, kwargs["T"], kwargs.get("verbose", False))
        if out is not None:
            return out
        noisy, clean = self._noisy_clean_pairs(x, kwargs["T"])
        recon = self.net.forward(x=noisy)
        batch_loss = self.loss(recon, clean)
        mean = batch_loss.mean()
        mean.backward()
        if kwargs.get("verbose", False):
            return batch_loss.abs(), recon.abs()
        return (mean.abs(),)

    def run_training_step_noise(self, x: torch.Tensor, **kwargs):
        out = self._fused_training_step(x, kwargs["T"], kwargs.get("verbose", False))
        if out is not None:
            return out
        noisy, clean = self._noisy_clean_pairs(x, kwargs["T"])
        predicted = (self.net.forward(x=noisy) - 0.5) * 0.1
        batch_loss = self.loss(predicted, noisy - clean)
        mean = batch_loss.mean()
        mean.backward()
        if kwargs.get("verbose", False):
            return batch_loss, torch.clamp(noisy - predicted, 0, 1)
        return (mean,)

    # -- sampling ------------------------------------------------------------------
    def denoise_step(self, x, noise_factor=1.0):
        """One body of the sampling loop (reference src/models.py:127-134): the unit behind
        BASELINE.json's "denoise-step images/sec"."""
        predicted = self.net(x)
        if self.prediction_goal == "data":
            return predicted
        return torch.clamp(x - (predicted - 0.5) * 0.1 * noise_factor, 0, 1)

    def denoise_steps(self, x, n, noise_factor=1.0):
        """``n`` consecutive loop bodies; returns the (n, *x.shape) stack of the images after each one.
        Nets that own a fused sampler (``fused_sample_steps``) run all n in one launch."""
        fused = getattr(self.net, "fused_sample_steps", None)
        if fused is not None and not torch.is_grad_enabled() and n > 0:
            out = fused(x, n, self.prediction_goal, noise_factor)
            if out is not None:
                return out
        outs, cur = [], x
        for _ in range(n):
            cur = self.denoise_step(cur, noise_factor)
            outs.append(cur)
        return torch.stack(outs) if outs else x.new_empty((0,) + tuple(x.shape))

    def sample(self, n_iters, first_x=None, labels=None, show_progress: bool = False,
               only_last=False, step=1, noise_factor=1.0) -> torch.Tensor:
        if first_x is None:
            first_x = torch.rand((10, 1, self.width, self.height))
        outp = [first_x]
        iters = range(n_iters)
        if show_progress and tqdm is not None:
            iters = tqdm.tqdm(iters)
        with torch.no_grad():
            if not (show_progress and tqdm is not None):
                steps = self.denoise_steps(first_x, n_iters, noise_factor)
                outp += [steps[i] for i in range(n_iters) if i % step == 0]
            else:
                x = first_x
                for i in iters:
                    x = self.denoise_step(x, noise_factor)
                    if i % step == 0:
                        outp.append(x)
        if only_last:
            return outp[-1]
        st = torch.stack(outp)                                  # iters batch 1 height width
        it, b, _, h, w = st.shape
        return st[:, :, 0].permute(0, 2, 1, 3).reshape(it * h, b * w)   # (iters height) (batch width)

    def save_name(self):
        return f"{self.net.save_name()}{'_noise' if self.prediction_goal == 'noise' else ''}"
