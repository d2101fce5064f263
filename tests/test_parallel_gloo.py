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
