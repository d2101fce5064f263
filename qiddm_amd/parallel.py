"""Data-parallel training glue: one process per GPU, `torch.distributed` (backend "nccl" is
RCCL on ROCm; "gloo" in the CPU tests).

The reference has no distributed code at all; the denoise path shards by
sample (section 8e), so the only exchange a training step needs is the gradient average.  The
payload is tiny (QNN_noise(784,8,14): 13 672 parameters = 107 KiB float64), i.e. latency-bound: all
gradients are flattened into ONE bucket and reduced with ONE collective per step, issued between
`Diffusion.forward` (which runs `.backward()` internally, reference src/models.py:67) and
`optimizer.step()`.
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of `n_items` for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x: torch.Tensor, rank: int | None = None, world: int | None = None) -> torch.Tensor:
    """This rank's contiguous slice of a global batch (dim 0)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Make every rank start from rank `src`'s weights (one flat bucket per dtype)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    _for_each_bucket([p.data for p in module.parameters()], lambda flat: dist.broadcast(flat, src))
    _for_each_bucket([b.data for b in module.buffers() if b.is_floating_point()],
                     lambda flat: dist.broadcast(flat, src))


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True) -> int:
    """Sum (or average) `.grad` over all ranks with ONE flat-bucket all-reduce per dtype.
    Parameters whose grad is None on this rank (e.g. the detached quantum weights, finding F1) are
    skipped -- they are None on every rank by construction.  Returns the number of elements reduced."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None]

    def reduce(flat):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat.div_(world)

    return _for_each_bucket(grads, reduce)


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


def training_step(diff, optimizer, x_local: torch.Tensor, T: int, verbose: bool = False):
    """One data-parallel step of the reference's hot loop 1 (src/mnist_exm.py:179-182):
    zero_grad -> diff(x, T) [forward + backward] -> gradient all-reduce -> optimizer.step()."""
    optimizer.zero_grad()
    out = diff(x=x_local, T=T, verbose=verbose)
    all_reduce_gradients(diff.parameters())
    optimizer.step()
    return out
