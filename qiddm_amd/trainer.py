"""Device-resident training step of the denoise loop.

The reference step is ``opt.zero_grad(); diff(x, T, verbose); opt.step()`` (src/mnist_exm.py:179-182) with
the noising, the ``(batch tau)`` re-indexing, the MSE and ``.backward()`` inside ``Diffusion``
(src/models.py:44-104).  Run eagerly that is ~60 small launches plus a host-side normal draw per step, and at
the benchmark shapes the launches, not the arithmetic, set the step time.  ``GraphedTrainStep`` records the
whole step -- noising, net forward (HIP circuit kernels), loss, backward (adjoint / parameter-shift kernels),
Adam -- once into HIP graphs and replays them:

    step = GraphedTrainStep(diff, FusedAdam(diff.parameters(), lr=...), x_example, T=10)   # or torch's Adam
                                                                                           # with capturable=True
    loss = step(x)            # same numbers as the eager step on the same noise

* ``noise="reference"``: the N(0.5, 0.2) field is drawn float32 on the CPU generator exactly as
  src/noise.py:113-115 does, then copied into the graph's static buffer (RNG-stream parity with an eager run).
* ``noise="device"``: drawn by the device generator inside the graph (no host work per step; a different,
  equally distributed stream).
* ``noise="fused"``: generated inside the fused training step itself (Philox4x32-10 + Box-Muller per element,
  seeded from ``torch.initial_seed()``): no RNG launches at all.  Only for nets with a fused training step.
* world size > 1: every ``.grad`` is a view into one persistent flat buffer (``parallel.GradBucket``) and the
  gradient all-reduce -- the one exchange step of the path (section 8e) -- is recorded INSIDE the graph between
  backward and Adam (RCCL collectives are capturable); with a backend that cannot be captured (gloo) the step is
  two graphs with the eager collective between them.  ``dp_weight`` = local_n / global_n for uneven shards
  (default 1 / world); ``shard=(global_n, lo, hi)`` makes ``noise="reference"`` draw the single-process field for
  the global batch and slice it (``parallel.ShardedNoise``).  With ``noise="fused"`` the Philox key mixes the rank in, so shards do not share a field.

Nets with a host-side front-end (the PCA classes, finding F4) cannot be recorded; they raise at capture and
the caller keeps the eager step.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import noise as _noise
from . import parallel


class _StaticNoise:
    """``add_normal_noise_multiple`` reading its one normal draw from the recorder's static buffer; carries the
    same ``noise_field`` / ``schedule`` markers, so a net's fused training step stays available."""

    schedule = staticmethod(_noise.add_normal_noise_multiple.schedule)

    def __init__(self, owner):
        self.owner = owner
        self.rng_state = owner.rng_state     # non-None: Diffusion hands it to the net's fused training step

    def noise_field(self, data):
        return self.owner._noise_field(data)

    def __call__(self, data, tau, decay_mod=1.0):
        if self.owner.noise_mode == "fused":     # the net declined its fused step: draw with the device generator
            self.owner.noise.normal_(mean=0.5, std=0.2)
        return _noise.add_normal_noise_multiple(data, tau, decay_mod, noise=self.owner._noise_field(data))


class GraphedTrainStep:
    def __init__(self, diff, optimizer, x_example, T=10, noise="reference", verbose=False, warmup=3,
                 dp_weight=None, shard=None, force_dp=False):
        if noise not in ("reference", "device", "fused"):
            raise ValueError(f"noise must be 'reference', 'device' or 'fused', got {noise!r}")
        if noise == "fused" and getattr(diff.net, "fused_train_step", None) is None:
            raise ValueError("noise='fused' needs a net with a fused training step")
        if not x_example.is_cuda:
            raise RuntimeError("GraphedTrainStep records HIP graphs: the batch must live on the GPU")
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("construct the optimizer with capturable=True")
        self.diff, self.opt, self.T, self.noise_mode, self.verbose = diff, optimizer, T, noise, verbose
        self.shard = shard          # (global_n, lo, hi): this rank's rows of the global batch, for noise="reference"
        if shard is not None and dp_weight is None:
            dp_weight = (shard[2] - shard[1]) / max(shard[0], 1)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.x = x_example.detach().clone()
        self.noise = torch.empty(self.x.shape, dtype=torch.float32, device=self.x.device)
        self.rng_state = None
        if noise == "fused":
            rank = dist.get_rank() if dist.is_initialized() else 0
            seed = (torch.initial_seed() + 0x9E3779B97F4A7C15 * rank) & ((1 << 63) - 1)     # per-rank Philox key
            self.rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=self.x.device)
            self.noise.fill_(0.5)
        self._draw_noise()
        self._user_noise_f = diff.add_noise
        params = [p for p in diff.parameters() if p.requires_grad]
        saved_p = [p.detach().clone() for p in params]
        buffers = list(diff.buffers())               # e.g. BatchNorm running statistics: the warm-up must not move them
        saved_b = [b.detach().clone() for b in buffers]
        # -- warm-up on a side stream (lazy workspaces, Adam state), then restore the initial parameters ----
        # A capturable torch optimizer keeps its step counter as a device scalar of the *default* dtype and
        # derives the bias corrections from it; with float32 that costs ~1e-5 relative in the first updates
        # against the eager optimizer (which uses Python floats).  The model is float64 (finding F5), so the
        # optimizer state is created under a float64 default.
        all_f64 = all(p.dtype == torch.float64 for p in params)
        prev_default = torch.get_default_dtype()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            if all_f64:
                torch.set_default_dtype(torch.float64)
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self.opt.zero_grad(set_to_none=True)
                    self._fwd_bwd()
                    self.opt.step()
        finally:
            torch.set_default_dtype(prev_default)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, s in zip(params, saved_p):
                p.copy_(s)
            for b, s in zip(buffers, saved_b):
                b.copy_(s)
            if hasattr(self.opt, "reset_state"):
                self.opt.reset_state()
            else:
                for st in self.opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            v.zero_()
        # -- record -------------------------------------------------------------------------------------
        self.g_fwd_bwd = torch.cuda.CUDAGraph()
        self.g_opt = None
        self.bucket = None
        self._params = params
        self.force_dp = bool(force_dp) and dist.is_initialized()    # one-rank group: still record the data-parallel step
        if self.world == 1 and not self.force_dp:
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.g_fwd_bwd):
                self._result = self._fwd_bwd()
                self.opt.step()
        else:
            # the warm-up left a gradient on every parameter the step trains: move them into ONE persistent flat
            # buffer (views), so the exchange is a single all-reduce of fixed addresses
            self.bucket = parallel.GradBucket.for_step(params)
            self.dp_weight = dp_weight
            # one eager exchange first: the communicator (RCCL: connections, channels, scratch) must exist before a
            # collective can be recorded into a graph; the values are discarded (the recorded step zeroes the bucket)
            self.bucket.all_reduce(self.dp_weight, force=self.force_dp)
            torch.cuda.synchronize()
            fused = dist.get_backend() == "nccl" and os.environ.get("QIDDM_DP_GRAPH", "fused") != "split"
            if fused:
                # RCCL collectives are capturable: zero -> forward+backward -> all-reduce -> Adam is ONE graph
                failure = None
                try:
                    with torch.cuda.graph(self.g_fwd_bwd):
                        self.bucket.zero()
                        self._result = self._fwd_bwd()
                        self.bucket.all_reduce(self.dp_weight, force=self.force_dp)
                        self.opt.step()
                except Exception as e:     # pragma: no cover - depends on the RCCL build
                    failure = e
                # the ranks must replay the SAME form (a rank with the collective inside its graph and a rank that
                # calls it eagerly would not pair up): agree on the minimum of the success flags
                torch.cuda.synchronize()
                ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32, device=self.x.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:    # pragma: no cover - depends on the RCCL build
                    import warnings
                    warnings.warn(f"capturing the gradient all-reduce failed on some rank ({failure!r} here); every "
                                  f"rank records two graphs around an eager all-reduce instead")
                    fused = False
                    # a capture that threw may have left its stream's capture state invalid: drop the graph object,
                    # drain the device, and record again from a clean state
                    self.g_fwd_bwd = None
                    torch.cuda.synchronize()
                    self.g_fwd_bwd = torch.cuda.CUDAGraph()
            if not fused:
                with torch.cuda.graph(self.g_fwd_bwd):
                    self.bucket.zero()
                    self._result = self._fwd_bwd()
                self.g_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_opt, pool=self.g_fwd_bwd.pool()):
                    self.opt.step()
            self.bucket.check_views()
        if self.rng_state is not None:
            self.rng_state[1] = 0               # the warm-up and the recording advanced the offset

    def _draw_noise(self):
        if self.noise_mode == "reference":
            if self.shard is None:
                field = torch.normal(mean=0.5, std=0.2, size=tuple(self.x.shape))
            else:       # the single-process draw for the global batch, sliced (parallel.ShardedNoise)
                n, lo, hi = self.shard
                field = torch.normal(mean=0.5, std=0.2, size=(n,) + tuple(self.x.shape[1:]))[lo:hi]
            self.noise.copy_(field, non_blocking=True)

    def _noise_field(self, data):
        if self.noise_mode == "device":
            self.noise.normal_(mean=0.5, std=0.2)
        return self.noise

    def _fwd_bwd(self):
        self.diff.add_noise = _StaticNoise(self)
        try:
            return self.diff(x=self.x, T=self.T, verbose=self.verbose)
        finally:
            self.diff.add_noise = self._user_noise_f

    def __call__(self, x):
        """One optimizer step on ``x`` (same shape as the example batch).  Returns what ``diff(x, T, verbose)``
        returns; the tensors are the graph's static outputs (overwritten by the next call)."""
        if x.shape != self.x.shape:
            raise ValueError(f"recorded for batch shape {tuple(self.x.shape)}, got {tuple(x.shape)}")
        self.x.copy_(x, non_blocking=True)
        self._draw_noise()
        self.g_fwd_bwd.replay()
        if self.g_opt is not None:
            self.bucket.all_reduce(self.dp_weight, force=self.force_dp)
            self.g_opt.replay()
        # a replay changes the parameters without any Python-side in-place op: tell torch's version counters, which
        # key the layers' caches of derived data (sampler tables, eval-mode circuit unitaries)
        for p in self._params:
            torch.autograd.graph.increment_version(p)
        return self._result
