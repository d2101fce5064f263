"""Recorded (HIP-graph) training step == the eager reference-order step on the same noise
(SURVEY.md section 8f rank 1)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(detach, method, seed=7):
    from qiddm_amd import models, nn, noise
    torch.manual_seed(seed)
    net = nn.QNN_noise(64, 4, 2, detach_quantum=detach)
    net.qnode.diff_method = method
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8),
                            torch.nn.MSELoss()).to("cuda", dtype=torch.double).train()
    return diff


@pytest.mark.parametrize("detach,method", [(True, "parameter-shift"), (False, "adjoint"), (False, "parameter-shift")])
def test_graphed_step_matches_eager(detach, method):
    from qiddm_amd.trainer import GraphedTrainStep
    xs = [torch.rand(6, 64, dtype=torch.double, device="cuda") for _ in range(3)]
    # eager, the reference's order of calls
    eager = _make(detach, method)
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-2)
    torch.manual_seed(123)
    losses_e = []
    for x in xs:
        opt_e.zero_grad()
        (loss,) = eager(x=x, T=5)
        opt_e.step()
        losses_e.append(loss.item())
    # recorded
    rec = _make(detach, method)
    opt_r = torch.optim.Adam(rec.parameters(), lr=1e-2, capturable=True)
    torch.manual_seed(99)          # the capture draws from the CPU generator too
    step = GraphedTrainStep(rec, opt_r, xs[0], T=5, noise="reference")
    torch.manual_seed(123)
    losses_r = [step(x)[0].item() for x in xs]
    assert losses_r == pytest.approx(losses_e, rel=1e-9, abs=1e-12)
    for (k, a), (_, b) in zip(eager.state_dict().items(), rec.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-7, atol=1e-10), k
    if not detach:
        assert not torch.equal(rec.net.weights, _make(detach, method).net.weights)   # the angles trained


def test_graphed_unet_simple_step_matches_eager():
    """The unet_simple training step (fused QConv2d forward / adjoint backward launches, BatchNorm in training mode,
    matrix-product glue) records into HIP graphs and replays to the eager numbers, BatchNorm statistics included."""
    from qiddm_amd import models, nn, noise
    from qiddm_amd.optim import FusedAdam
    from qiddm_amd.trainer import GraphedTrainStep

    def make():
        torch.manual_seed(3)
        net = nn.UNetUndirectedS(2, 4, 2)
        return models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8),
                                torch.nn.MSELoss()).to("cuda", dtype=torch.double).train()

    xs = [torch.rand(3, 64, dtype=torch.double, device="cuda") for _ in range(3)]
    eager = make()
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-2)
    torch.manual_seed(123)
    losses_e = []
    for x in xs:
        opt_e.zero_grad()
        (loss,) = eager(x=x, T=4)
        opt_e.step()
        losses_e.append(loss.item())
    rec = make()
    step = GraphedTrainStep(rec, FusedAdam(rec.parameters(), lr=1e-2), xs[0], T=4, noise="reference")
    torch.manual_seed(123)
    losses_r = [step(x)[0].item() for x in xs]
    assert losses_r == pytest.approx(losses_e, rel=1e-6, abs=1e-9)
    for (k, a), (_, b) in zip(eager.state_dict().items(), rec.state_dict().items()):
        assert torch.allclose(a.double(), b.double(), rtol=1e-5, atol=1e-7), k


def test_device_noise_and_shape_check():
    from qiddm_amd.trainer import GraphedTrainStep
    diff = _make(True, "parameter-shift")
    opt = torch.optim.Adam(diff.parameters(), lr=1e-2, capturable=True)
    x = torch.rand(4, 64, dtype=torch.double, device="cuda")
    step = GraphedTrainStep(diff, opt, x, T=5, noise="device")
    l0 = step(x)[0].item()
    l1 = step(x)[0].item()
    assert l0 > 0 and l1 > 0 and l0 != l1
    with pytest.raises(ValueError):
        step(torch.rand(5, 64, dtype=torch.double, device="cuda"))
    with pytest.raises(ValueError):
        GraphedTrainStep(diff, torch.optim.Adam(diff.parameters(), lr=1e-2), x, T=5)


def test_fused_adam_matches_torch_adam():
    from qiddm_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(8, 784), (8,), (14, 8, 3), (784, 8), (784,)] + [(5,)] * 14       # > 16 tensors: two launches
    for dtype, tol in ((torch.float64, 1e-13), (torch.float32, 2e-6)):
        a = [torch.randn(s, dtype=dtype, device="cuda").requires_grad_(True) for s in shapes]
        b = [t.detach().clone().requires_grad_(True) for t in a]
        oa = torch.optim.Adam(a, lr=0.01011, weight_decay=0.01)
        ob = FusedAdam(b, lr=0.01011, weight_decay=0.01)
        for it in range(5):
            for ta, tb in zip(a, b):
                g = torch.randn(ta.shape, dtype=dtype, device="cuda") * (10.0 ** -it)
                ta.grad, tb.grad = g.clone(), g.clone()
            a[1].grad = b[1].grad = None if it == 2 else a[1].grad     # a parameter without gradient is skipped
            oa.step()
            ob.step()
        for ta, tb in zip(a, b):
            assert torch.allclose(ta, tb, rtol=tol, atol=tol), (dtype, ta.shape)
        assert set(ob.state[b[0]].keys()) == {"step", "exp_avg", "exp_avg_sq"}
        assert ob.state[b[0]]["step"].item() == 5 and ob.state[b[1]]["step"].item() == 4
    cpu = torch.zeros(3, requires_grad=True)
    cpu.grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        FusedAdam([cpu]).step()                       # no CPU path


def test_graphed_step_with_fused_adam_matches_eager_torch_adam():
    from qiddm_amd.optim import FusedAdam
    from qiddm_amd.trainer import GraphedTrainStep
    xs = [torch.rand(6, 64, dtype=torch.double, device="cuda") for _ in range(4)]
    eager = _make(False, "adjoint")
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-2)
    torch.manual_seed(123)
    losses_e = []
    for x in xs:
        opt_e.zero_grad()
        (loss,) = eager(x=x, T=5)
        opt_e.step()
        losses_e.append(loss.item())
    rec = _make(False, "adjoint")
    step = GraphedTrainStep(rec, FusedAdam(rec.parameters(), lr=1e-2), xs[0], T=5, noise="reference")
    torch.manual_seed(123)
    losses_r = [step(x)[0].item() for x in xs]
    assert losses_r == pytest.approx(losses_e, rel=1e-9, abs=1e-12)
    for (k, a), (_, b) in zip(eager.state_dict().items(), rec.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-7, atol=1e-10), k


def test_fused_noise_generation():
    """noise="fused": the N(0.5, 0.2) field is generated inside the training launch (Philox) and recorded in the
    step's buffer -- the loss must be the eager loss on exactly that field, and every replay draws a new one."""
    from oracle import diffusion as odf
    from qiddm_amd.optim import FusedAdam
    from qiddm_amd.trainer import GraphedTrainStep
    torch.manual_seed(5)
    diff = _make(False, "adjoint")
    x = torch.rand(64, 64, dtype=torch.double, device="cuda")
    step = GraphedTrainStep(diff, FusedAdam(diff.parameters(), lr=0.0), x, T=5, noise="fused")   # lr 0: weights fixed
    fields, losses = [], []
    for _ in range(3):
        losses.append(step(x)[0].item())
        fields.append(step.noise.clone())
    assert not torch.equal(fields[0], fields[1]) and not torch.equal(fields[1], fields[2])
    allf = torch.cat([f.flatten() for f in fields]).double()
    assert abs(allf.mean().item() - 0.5) < 0.01 and abs(allf.std().item() - 0.2) < 0.01
    assert abs(((allf - 0.5) / 0.2).pow(3).mean().item()) < 0.1 and abs(((allf - 0.5) / 0.2).pow(4).mean().item() - 3.0) < 0.2
    # same numbers as the eager step on the recorded field
    diff.net.fused_train_step = None
    for f, want in zip(fields, losses):
        noisy, clean = odf.training_pairs(x.cpu(), 5, (8, 8), noise=f.cpu())
        with torch.no_grad():
            out = diff.net(noisy.to("cuda"))
        assert ((out.cpu() - clean) ** 2).mean().item() == pytest.approx(want, rel=1e-5)


def test_graphed_step_with_unet_simple_and_batchnorm_buffers():
    """A net outside the fused step (unet_simple: QConv2d through the adjoint kernel, BatchNorm in training mode):
    the recorded step reproduces the eager one, including the running statistics."""
    from qiddm_amd import models, nn, noise
    from qiddm_amd.trainer import GraphedTrainStep

    def make():
        torch.manual_seed(3)
        net = nn.UNetUndirectedS(2, 2, 1)
        return models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8),
                                torch.nn.MSELoss()).to("cuda", dtype=torch.double).train()
    xs = [torch.rand(3, 64, dtype=torch.double, device="cuda") for _ in range(2)]
    eager = make()
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-2)
    torch.manual_seed(77)
    le = []
    for x in xs:
        opt_e.zero_grad()
        (loss,) = eager(x=x, T=2)
        opt_e.step()
        le.append(loss.item())
    rec = make()
    step = GraphedTrainStep(rec, torch.optim.Adam(rec.parameters(), lr=1e-2, capturable=True), xs[0], T=2,
                            noise="reference")
    torch.manual_seed(77)
    lr_ = [step(x)[0].item() for x in xs]
    assert lr_ == pytest.approx(le, rel=1e-7)
    for (k, a), (_, b) in zip(eager.state_dict().items(), rec.state_dict().items()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-9), k
        else:
            assert torch.equal(a, b), k           # num_batches_tracked


def test_recorded_data_parallel_step_with_rccl_in_the_graph():
    """The data-parallel form of the recorded step on the one GPU of the box: a ONE-rank RCCL ("nccl") group, every
    ``.grad`` a view into the flat bucket, bucket.zero -> forward+backward -> all_reduce -> Adam recorded as ONE HIP
    graph (``force_dp``).  With one rank the collective is the identity, so the numbers must equal the eager loop --
    what is under test is that the RCCL call records and replays, that the fused training step and the one-launch Adam
    work on bucket views, and that nothing detaches them."""
    import socket
    import torch.distributed as dist
    from qiddm_amd.optim import FusedAdam
    from qiddm_amd.trainer import GraphedTrainStep
    xs = [torch.rand(6, 64, dtype=torch.double, device="cuda") for _ in range(3)]
    eager = _make(False, "adjoint")
    opt_e = torch.optim.Adam(eager.parameters(), lr=1e-2)
    torch.manual_seed(123)
    losses_e = []
    for x in xs:
        opt_e.zero_grad()
        (loss,) = eager(x=x, T=5)
        opt_e.step()
        losses_e.append(loss.item())
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        rec = _make(False, "adjoint")
        torch.manual_seed(99)
        step = GraphedTrainStep(rec, FusedAdam(rec.parameters(), lr=1e-2), xs[0], T=5, noise="reference", force_dp=True)
        assert step.bucket is not None and step.g_opt is None          # one graph, the collective inside it
        assert step.bucket.numel() == sum(p.numel() for p in rec.parameters())
        torch.manual_seed(123)
        losses_r = [step(x)[0].item() for x in xs]
        step.bucket.check_views()
    finally:
        dist.destroy_process_group()
    assert losses_r == pytest.approx(losses_e, rel=1e-9, abs=1e-12)
    for (k, a), (_, b) in zip(eager.state_dict().items(), rec.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-7, atol=1e-10), k


def test_weight_gradient_side_stream_leaves_the_same_gradients(monkeypatch):
    """The convolutions' weight-gradient chains run on a side stream (circuit.on_weight_grad_stream): the same kernels on
    the same data, so every gradient equals the single-stream run bit for bit -- over several steps back to back, so
    that a missing join between the side stream and the optimizer (or the next forward) would show."""
    from qiddm_amd import circuit, models, nn, noise

    def run(side):
        monkeypatch.setattr(circuit, "_WEIGHT_GRAD_STREAM", side)
        torch.manual_seed(5)
        net = nn.UNetUndirectedS(2, 4, 2)
        diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8),
                                torch.nn.MSELoss()).to("cuda", dtype=torch.double).train()
        opt = torch.optim.Adam(diff.parameters(), lr=1e-2)
        g = torch.Generator(device="cuda").manual_seed(9)
        grads, losses = [], []
        torch.manual_seed(11)
        for _ in range(4):
            x = torch.rand(6, 64, dtype=torch.double, device="cuda", generator=g)
            opt.zero_grad()
            (loss,) = diff(x=x, T=3)
            grads.append({k: p.grad.clone() for k, p in diff.named_parameters() if p.grad is not None})
            opt.step()
            losses.append(loss.item())
        return grads, losses, {k: v.clone() for k, v in diff.state_dict().items()}

    g1, l1, s1 = run(True)
    assert circuit.weight_grad_stream(torch.device("cuda", 0)) is not None
    g0, l0, s0 = run(False)
    assert circuit.weight_grad_stream(torch.device("cuda", 0)) is None
    assert l1 == l0
    for a, b in zip(g1, g0):
        assert a.keys() == b.keys() and any("weights" in k for k in a)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for k in s1:
        assert torch.equal(s1[k], s0[k]), k


@pytest.mark.parametrize("c_in,c_out,k,pad,hw", [(1, 8, 3, 1, (28, 28)), (16, 8, 3, 1, (28, 28)), (32, 16, 1, 0, (14, 14)),
                                                  (16, 32, 3, 1, (7, 7)), (5, 3, 3, 0, (9, 8))])
def test_qconv_batchnorm_pair_as_one_node_matches_the_two_modules(c_in, c_out, k, pad, hw, monkeypatch):
    """[QConv2d, BatchNorm2d] in training mode as one autograd node (the BatchNorm backward's transform folded into the
    convolution's thin-product kernel) against the two modules run one after the other: same output, running statistics
    and gradients of the input, the circuit weights and the BatchNorm affine pair.  Shapes: the VALU kernel (first layer
    of unet_simple), the matrix-core kernel with and without the per-pixel-row dL/dx, and a layer without padding."""
    from qiddm_amd import circuit, nn
    from qiddm_amd.nn.unet import _run

    def run(fused):
        monkeypatch.setattr(circuit, "_QCONV_BN_FUSED", fused)
        torch.manual_seed(17)
        net = torch.nn.Sequential(nn.QConv2d(c_in, c_out, k, pad, 2), torch.nn.BatchNorm2d(c_out)).to("cuda", torch.double).train()
        with torch.no_grad():
            net[1].weight.uniform_(0.5, 1.5)
            net[1].bias.uniform_(-0.5, 0.5)
        x = torch.rand(4, c_in, *hw, dtype=torch.double, device="cuda").requires_grad_(True)
        out = _run(net, x)
        g = torch.randn(out.shape, dtype=torch.double, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
        (out * g).sum().backward()
        return out.detach(), x.grad, {n: p.grad for n, p in net.named_parameters()}, \
            {n: b.clone() for n, b in net.named_buffers()}, type(out.grad_fn).__name__

    out1, gx1, gp1, buf1, node1 = run(True)
    out0, gx0, gp0, buf0, node0 = run(False)
    # (the matrix-core kernel with 32 row channels has no registers left for the fold: that layer keeps two nodes)
    foldable = circuit.qconv_bn_foldable((4, c_in) + tuple(hw), nn.QConv2d(c_in, c_out, k, pad, 2).wires, c_out, (k, k), (pad, pad))
    assert foldable == (c_out <= 16)
    assert node1 == ("_QConvBNTrainFunctionBackward" if foldable else "_BatchNormTrainFunctionBackward")
    assert node0 == "_BatchNormTrainFunctionBackward"
    assert torch.equal(out1, out0)
    for n in buf1:
        assert torch.equal(buf1[n], buf0[n]), n
    scale = max(1.0, gx0.abs().max().item())
    assert torch.allclose(gx1, gx0, rtol=0, atol=1e-5 * scale), (gx1 - gx0).abs().max()
    for n in gp1:
        s = max(1.0, gp0[n].abs().max().item())
        assert torch.allclose(gp1[n], gp0[n], rtol=0, atol=1e-5 * s), (n, (gp1[n] - gp0[n]).abs().max())
