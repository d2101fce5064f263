"""N > 1 path on CPU: world_size-2 gloo processes.  The quantum layers need a GPU, so the net here
is a classical stand-in; what is under test is the sharding + single flat-bucket gradient
all-reduce + `Diffusion` training-step wiring."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(64, 64, dtype=torch.double)
        self.frozen = torch.nn.Parameter(torch.ones(3, dtype=torch.double))   # grad stays None (cf. F1)
        self.f32p = torch.nn.Parameter(torch.zeros(5, dtype=torch.float32))   # second dtype bucket

    def forward(self, x):
        b = x.shape[0]
        return torch.sigmoid(self.lin(x.reshape(b, -1))).reshape(x.shape) + self.f32p.sum().double()

    def save_name(self):
        return "tiny"


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qiddm_amd import models, parallel
        torch.manual_seed(100 + rank)                      # ranks start different ...
        net = TinyNet()
        parallel.broadcast_parameters(net, src=0)          # ... and are made identical
        diff = models.Diffusion(net, lambda d, tau, decay_mod: _fixed_noise(d, tau, decay_mod), "data", (8, 8),
                                torch.nn.MSELoss())
        diff.train()
        opt = torch.optim.SGD(diff.parameters(), lr=0.5)
        torch.manual_seed(7)
        x_global = torch.rand(6, 64, dtype=torch.double)
        x_local = parallel.shard_batch(x_global)
        parallel.training_step(diff, opt, x_local, T=4)
        ret[rank] = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ret[f"n{rank}"] = x_local.shape[0]
    finally:
        dist.destroy_process_group()


def _fixed_noise(data, tau, decay_mod):
    """Deterministic stand-in for add_normal_noise_multiple so both world sizes see the same data."""
    w = (torch.linspace(0, 1, tau, dtype=data.dtype) ** decay_mod).reshape(1, tau, 1)
    noisy = data.unsqueeze(1) * (1 - w) + 0.5 * w
    return noisy.reshape(data.shape[0] * tau, -1)


def test_shard_bounds_cover_everything():
    from qiddm_amd.parallel import shard_bounds
    for n in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.timeout(180)
def test_dp2_matches_single_process():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["n0"] == ret["n1"] == 3
    for k in ret[0]:
        assert torch.equal(ret[0][k], ret[1][k]), k                      # ranks stay in lock-step
    # single-process reference on the full batch (loss = mean over the batch => grads average)
    from qiddm_amd import models
    torch.manual_seed(100)
    net = TinyNet()
    diff = models.Diffusion(net, _fixed_noise, "data", (8, 8), torch.nn.MSELoss())
    diff.train()
    opt = torch.optim.SGD(diff.parameters(), lr=0.5)
    torch.manual_seed(7)
    x_global = torch.rand(6, 64, dtype=torch.double)
    opt.zero_grad()
    diff(x=x_global, T=4)
    opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(ret[0][k], v, atol=1e-12), k
    assert torch.equal(ret[0]["frozen"], torch.ones(3, dtype=torch.double))


# ---- the flat-bucket path: uneven / empty shards, the real noise schedule ---------------------------------------
def _dp_worker(rank, world, port, ret, batch_sizes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qiddm_amd import models, noise, parallel
        torch.manual_seed(100)
        net = TinyNet()
        parallel.broadcast_parameters(net, src=0)
        diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8), torch.nn.MSELoss())
        diff.train()
        opt = torch.optim.Adam(diff.parameters(), lr=0.05)
        step = parallel.DataParallelStep(diff, opt)
        torch.manual_seed(7)                                # every rank: the same global batches and noise stream
        for n in batch_sizes:
            x_global = torch.rand(n, 64, dtype=torch.double)
            step(x_global, T=4)
        step.bucket.check_views()
        ret[rank] = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ret[f"rng{rank}"] = torch.get_rng_state()
        ret[f"bucket{rank}"] = (step.bucket.numel(), len(step.bucket.flats), net.frozen.grad is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("batch_sizes", [(5, 1, 6), (1, 3)])
def test_dp2_uneven_and_empty_shards_match_single_process(batch_sizes):
    """5 = 3 + 2 (uneven), 1 = 1 + 0 (a rank with an EMPTY shard joins with zeros), real
    ``add_normal_noise_multiple`` (global draw, sliced): DP-2 == the single-process run, generators in lock-step."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_dp_worker, args=(world, port, ret, batch_sizes), nprocs=world, join=True)
    for k in ret[0]:
        assert torch.equal(ret[0][k], ret[1][k]), k
    assert torch.equal(ret["rng0"], ret["rng1"])
    # members: lin.weight, lin.bias (float64) + f32p (float32) -> two flat buffers; `frozen` keeps grad None
    assert ret["bucket0"] == ret["bucket1"] == (64 * 64 + 64 + 5, 2, True)
    from qiddm_amd import models, noise
    torch.manual_seed(100)
    net = TinyNet()
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8), torch.nn.MSELoss())
    diff.train()
    opt = torch.optim.Adam(diff.parameters(), lr=0.05)
    torch.manual_seed(7)
    for n in batch_sizes:
        x_global = torch.rand(n, 64, dtype=torch.double)
        opt.zero_grad()
        diff(x=x_global, T=4)
        opt.step()
    for k, v in net.state_dict().items():
        tol = 1e-12 if v.dtype == torch.float64 else 1e-6       # the float32 bucket rounds its weighted sum in float32
        assert torch.allclose(ret[0][k], v, atol=tol, rtol=1e-10), (k, (ret[0][k] - v).abs().max())
    assert torch.equal(ret["rng0"], torch.get_rng_state())


def test_grad_bucket_views_single_process():
    from qiddm_amd.parallel import GradBucket
    net = TinyNet()
    x = torch.rand(4, 64, dtype=torch.double)
    net(x).sum().backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    b = GradBucket.for_step(net.parameters())
    assert net.frozen.grad is None and b.numel() == 64 * 64 + 64 + 5
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, want[n])
    b.zero()
    net(x).sum().backward()                       # autograd accumulates into the views in place
    b.check_views()
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, want[n])
    torch.optim.SGD(net.parameters(), lr=0.1).zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError):
        b.check_views()


# ---- a noise function WITHOUT the noise_field marker: drawn for the global batch on every rank, rows sliced ----------
def _plain_random_noise(data, tau, decay_mod=1.0):
    """Not add_normal_noise_multiple: no `noise_field` / `schedule` attributes, draws from the default CPU generator."""
    field = torch.rand(data.shape[0], data.shape[1], dtype=data.dtype)
    w = (torch.linspace(0, 1, tau, dtype=data.dtype) ** decay_mod).reshape(1, tau, 1)
    return (data.unsqueeze(1) * (1 - w) + field.unsqueeze(1) * w).reshape(data.shape[0] * tau, -1)


def _dp_fallback_worker(rank, world, port, ret, batch_sizes):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from qiddm_amd import models, parallel
        torch.manual_seed(100)
        net = TinyNet()
        diff = models.Diffusion(net, _plain_random_noise, "data", (8, 8), torch.nn.MSELoss())
        diff.train()
        opt = torch.optim.Adam(diff.parameters(), lr=0.05)
        step = parallel.DataParallelStep(diff, opt)
        torch.manual_seed(7)
        for n in batch_sizes:
            step(torch.rand(n, 64, dtype=torch.double), T=4)
        ret[rank] = {k: v.detach().clone() for k, v in net.state_dict().items()}
        ret[f"rng{rank}"] = torch.get_rng_state()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_dp2_unmarked_noise_function_is_drawn_globally_and_sliced():
    """Round-2 advice: without `noise_field` every rank drew a LOCAL-size field from the same-seeded generator (identical
    noise on every rank, generators drifting apart on uneven shards).  Now: the global draw, sliced -> DP-2 equals the
    single-process run and the generators end equal, uneven (3 + 2) and empty (1 + 0) shards included."""
    world, batch_sizes = 2, (5, 1, 4)
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_dp_fallback_worker, args=(world, port, ret, batch_sizes), nprocs=world, join=True)
    assert torch.equal(ret["rng0"], ret["rng1"])
    from qiddm_amd import models
    torch.manual_seed(100)
    net = TinyNet()
    diff = models.Diffusion(net, _plain_random_noise, "data", (8, 8), torch.nn.MSELoss())
    diff.train()
    opt = torch.optim.Adam(diff.parameters(), lr=0.05)
    torch.manual_seed(7)
    for n in batch_sizes:
        x_global = torch.rand(n, 64, dtype=torch.double)
        opt.zero_grad()
        diff(x=x_global, T=4)
        opt.step()
    for k, v in net.state_dict().items():
        tol = 1e-12 if v.dtype == torch.float64 else 1e-6
        assert torch.allclose(ret[0][k], v, atol=tol, rtol=1e-10), (k, (ret[0][k] - v).abs().max())
        assert torch.equal(ret[0][k], ret[1][k]), k
    assert torch.equal(ret["rng0"], torch.get_rng_state())
