"""Data-parallel training glue: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; the denoise path shards by
sample (section 8e), so the only exchange a training step needs is the gradient average.  The
payload is tiny (QNN_noise(784,8,14): 13 672 parameters = 107 KiB float64), i.e. latency-bound, so
the glue around the collective is the cost:

* ``GradBucket``: every ``.grad`` is a *view* into ONE persistent flat buffer per (dtype, device).
  Backward passes (autograd and the fused HIP training step alike) accumulate into the views in
  place, the exchange is a single ``all_reduce`` of the buffer itself -- no ``cat``, no copy-back, no
  allocation per step -- and because the addresses never change the collective can sit inside a
  recorded HIP graph between backward and the optimizer (``qiddm_amd.trainer``).
* Shards may be uneven or empty (a global batch smaller than the world, an epoch's last batch): every
  rank scales its gradient by ``local_n / global_n`` before the SUM, which is the gradient of the
  global batch mean (the loss is a mean over samples, reference src/models.py:65-67); a rank with an
  empty shard skips forward/backward and joins the collective with zeros.
* ``ShardedNoise``: the one N(0.5, 0.2) draw of ``add_normal_noise_multiple`` (reference
  src/noise.py:113-115) is made for the GLOBAL batch on every rank -- same CPU generator, same
  seed, same order as the single-process run -- and sliced to the rank's shard, so the ranks' noise
  is the single-process field (not N copies of one field) and the generators stay in lock-step.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def _world() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def _rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `n_items` for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x: torch.Tensor, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """This rank's contiguous slice of a global batch (dim 0); may be empty."""
    lo, hi = shard_bounds(x.shape[0], _rank() if rank is None else rank, _world() if world is None else world)
    return x[lo:hi]


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s weights (one flat bucket per dtype)."""
    if _world() == 1:
        return
    _for_each_bucket([p.data for p in module.parameters()], lambda flat: dist.broadcast(flat, src))
    _for_each_bucket([b.data for b in module.buffers() if b.is_floating_point()],
                     lambda flat: dist.broadcast(flat, src))


def _for_each_bucket(tensors: List[torch.Tensor], fn) -> int:
    total = 0
    by_key = {}
    for t in tensors:
        by_key.setdefault((t.dtype, t.device), []).append(t)
    for group in by_key.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        fn(flat)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
        total += flat.numel()
    return total


class GradBucket:
    """Persistent flat gradient storage: ``p.grad`` of every member is a view into one buffer per
    (dtype, device).

    ``members`` are the parameters that receive a gradient in a training step (with the reference's
    detached quantum weights, finding F1, that is ``linear_up`` only: the others keep ``.grad is None``
    exactly as after the reference's ``.backward()``, and ``Adam`` skips them).  Use ``for_step`` to let a
    first backward decide (the union over ranks, so a rank with an empty shard still joins with zeros).
    """

    def __init__(self, members: Iterable[torch.nn.Parameter]):
        self.members = [p for p in members if p.requires_grad]
        self.flats = {}                 # (dtype, device) -> flat buffer
        by_key = {}
        for p in self.members:
            by_key.setdefault((p.dtype, p.device), []).append(p)
        for key, group in by_key.items():
            flat = torch.zeros(sum(p.numel() for p in group), dtype=key[0], device=key[1])
            off = 0
            for p in group:
                n = p.numel()
                view = flat[off:off + n].view(p.shape)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
                off += n
            self.flats[key] = flat

    @classmethod
    def for_step(cls, params: Iterable[torch.nn.Parameter]) -> "GradBucket":
        """Adopt the gradients a backward pass has just left: members = parameters whose ``.grad`` is set on ANY
        rank (one tiny MAX all-reduce of the mask, once), current values are kept."""
        params = [p for p in params if p.requires_grad]
        mask = torch.tensor([0 if p.grad is None else 1 for p in params], dtype=torch.int32)
        if _world() > 1:
            dev = params[0].device if dist.get_backend() != "gloo" and params else torch.device("cpu")
            m = mask.to(dev)
            dist.all_reduce(m, op=dist.ReduceOp.MAX)
            mask = m.cpu()
        return cls([p for p, k in zip(params, mask.tolist()) if k])

    def numel(self) -> int:
        return sum(f.numel() for f in self.flats.values())

    def zero(self) -> None:
        """``optimizer.zero_grad()`` without dropping the views: one fill per dtype."""
        for flat in self.flats.values():
            flat.zero_()

    def check_views(self) -> None:
        """Raise if something (``zero_grad(set_to_none=True)``, ``p.grad = new``) detached a member from the buffer."""
        for p in self.members:
            flat = self.flats[(p.dtype, p.device)]
            lo = flat.data_ptr()
            if p.grad is None or not (lo <= p.grad.data_ptr() < lo + flat.numel() * flat.element_size()):
                raise RuntimeError("a parameter's .grad no longer aliases the flat bucket; use bucket.zero() instead "
                                   "of optimizer.zero_grad() and accumulate into .grad in place")

    def all_reduce(self, weight: float = None, force: bool = False) -> int:
        """The one exchange step: ``grad <- sum_ranks weight_r * grad_r`` in place, ONE collective per dtype.
        ``weight`` = local_n / global_n (default 1 / world: equal shards).  Capturable into a HIP graph (fixed
        addresses, no allocation).  Returns the number of elements reduced.  ``force``: issue the collective even in a
        one-rank group (how the single-GPU box exercises the recorded RCCL call)."""
        world = _world()
        if world == 1 and not (force and dist.is_initialized()):
            return 0
        w = (1.0 / world) if weight is None else float(weight)
        for flat in self.flats.values():
            if w != 1.0:
                flat.mul_(w)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return self.numel()


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True, weight: float = None) -> int:
    """Stateless form for callers without a ``GradBucket``: sum (or average / weight) `.grad` over all ranks with
    one flat-bucket all-reduce per dtype (a ``cat`` and a copy-back per call -- the bucket avoids both).
    Parameters whose grad is None on this rank are skipped -- they must be None on every rank."""
    if _world() == 1:
        return 0
    world = _world()
    grads = [p.grad for p in params if p.grad is not None]
    w = weight if weight is not None else ((1.0 / world) if average else 1.0)

    def reduce(flat):
        if w != 1.0:
            flat.mul_(w)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)

    return _for_each_bucket(grads, reduce)


class ShardedNoise:
    """``add_normal_noise_multiple`` for one rank's shard ``[lo, hi)`` of a global batch of ``global_n`` samples:
    the field is drawn for the global batch (float32, default CPU generator: reference src/noise.py:113-115) and
    sliced, so DP-N applies exactly the single-process noise and every rank's generator advances identically.
    Carries the ``noise_field`` / ``schedule`` markers, i.e. the fused training step stays available."""

    def __init__(self, base_noise_f, global_n: int, lo: int, hi: int):
        self.base, self.global_n, self.lo, self.hi = base_noise_f, int(global_n), int(lo), int(hi)
        self.schedule = getattr(base_noise_f, "schedule", None)

    def noise_field(self, data):
        full = torch.normal(mean=0.5, std=0.2, size=(self.global_n, data.shape[-1]))
        return full[self.lo:self.hi].to(data.device)

    def __call__(self, data, tau, decay_mod=1.0):
        return self.base(data, tau, decay_mod, noise=self.noise_field(data))


class GlobalSliceNoise:
    """Fallback for noise functions WITHOUT the ``noise_field`` marker (anything but ``add_normal_noise_multiple``):
    every rank runs the function on the GLOBAL batch -- so each draws what the single process would, and all generators
    stay in lock-step whatever the shard sizes -- and keeps the rows of its own samples (the output is batch-major,
    ``tau`` rows per sample: reference src/noise.py:121-126)."""

    def __init__(self, base_noise_f, x_global: torch.Tensor, lo: int, hi: int):
        self.base, self.x_global, self.lo, self.hi = base_noise_f, x_global, int(lo), int(hi)

    def __call__(self, data, tau, decay_mod=1.0):
        n = self.x_global.shape[0]
        whole = self.base(self.x_global.reshape(n, -1), tau, decay_mod)
        return whole.reshape(n, tau, -1)[self.lo:self.hi].reshape((self.hi - self.lo) * tau, -1)


class DataParallelStep:
    """One data-parallel step of the reference's hot loop 1 (src/mnist_exm.py:179-182):
    ``zero_grad -> diff(x_local, T) [forward + backward] -> gradient all-reduce -> optimizer.step()`` on this rank's
    shard of every global batch, robust to uneven and empty shards."""

    def __init__(self, diff, optimizer):
        self.diff, self.opt = diff, optimizer
        self.bucket: Optional[GradBucket] = None
        self._noise_f = diff.add_noise

    def __call__(self, x_global: torch.Tensor, T: int, verbose: bool = False):
        world, rank = _world(), _rank()
        n = x_global.shape[0]
        lo, hi = shard_bounds(n, rank, world)
        x_local = x_global[lo:hi]
        if self.bucket is None:
            self.opt.zero_grad(set_to_none=True)
        else:
            self.bucket.zero()
        out = None
        has_field = getattr(self._noise_f, "noise_field", None) is not None
        if world > 1:
            self.diff.add_noise = ShardedNoise(self._noise_f, n, lo, hi) if has_field else \
                GlobalSliceNoise(self._noise_f, x_global, lo, hi)
        try:
            if hi > lo:
                out = self.diff(x=x_local, T=T, verbose=verbose)
            elif world > 1:                                          # keep the generator in lock-step
                if has_field:
                    self.diff.add_noise.noise_field(x_global[:1])
                else:
                    self._noise_f(x_global.reshape(n, -1), T + 1, 3.0)   # what Diffusion._noisy_clean_pairs would draw
        finally:
            self.diff.add_noise = self._noise_f
        if self.bucket is None:
            self.bucket = GradBucket.for_step(self.diff.parameters())
        self.bucket.all_reduce(weight=(hi - lo) / max(n, 1))
        self.opt.step()
        return out


def training_step(diff, optimizer, x_local: torch.Tensor, T: int, verbose: bool = False):
    """Equal-shard form kept for callers that shard themselves: zero_grad -> diff(x_local, T) -> gradient
    all-reduce (mean over ranks) -> optimizer.step()."""
    optimizer.zero_grad()
    out = diff(x=x_local, T=T, verbose=verbose)
    all_reduce_gradients(diff.parameters())
    optimizer.step()
    return out
