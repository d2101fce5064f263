"""Row A6 end to end: ``UNetUndirectedS`` on the HIP path against ``oracle.unet.unet_simple_forward`` (quantum
convolutions through the statevector oracle, glue in CPU torch float64) -- eval mode (circuit unitary + MFMA GEMM,
fused bilinear x2 / BatchNorm epilogues, HIP 1x1 head) and train mode (batch statistics; unitary route in f32,
per-pixel circuit simulation in f64).  Reference: nn/unet_simple.py:6-84, nn/unet.py:70-75, 111-116, 162-174,
nn/utils.py:22-39."""
import pytest
import torch

from oracle import unet as ou

pytestmark = pytest.mark.gpu


def _net(depth, start, qdepth, seed):
    from qiddm_amd import nn
    torch.manual_seed(seed)
    net = nn.UNetUndirectedS(depth, start, qdepth)
    # non-trivial BatchNorm state: the fresh module (gamma 1, beta 0, mean 0, var 1) would hide an epilogue slip
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(0.0, 0.5)
                m.running_var.uniform_(0.05, 0.5)
    return net.to("cuda", dtype=torch.double)


@pytest.mark.parametrize("shape,depth,start", [((2, 1, 28, 28), 3, 8),      # C3's net at the reference resolution
                                               ((2, 1, 10, 10), 3, 8),      # 10 -> 5 -> 2 -> 4 vs 5: autopad path
                                               ((3, 1, 12, 12), 2, 4)])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_unet_simple_eval_matches_oracle(shape, depth, start, precision):
    import qiddm_amd
    net = _net(depth, start, 3, seed=5).eval()
    x = torch.rand(*shape, dtype=torch.double, generator=torch.Generator().manual_seed(6))
    qiddm_amd.set_default_precision(precision)
    try:
        with torch.no_grad():
            got = net(x.cuda()).cpu()
    finally:
        qiddm_amd.set_default_precision("f32")
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    want = ou.unet_simple_forward(x, sd, depth, start, training=False)
    assert got.shape == want.shape == shape
    tol = 5e-4 if precision == "f32" else 1e-9
    assert torch.allclose(got, want, atol=tol, rtol=tol), (got - want).abs().max()


@pytest.mark.parametrize("shape,depth,start", [((2, 1, 28, 28), 3, 8), ((4, 1, 10, 10), 3, 8)])
@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_unet_simple_train_mode_matches_oracle(shape, depth, start, precision):
    """Training-mode forward: batch statistics in every BatchNorm2d, running statistics moved as torch does."""
    import qiddm_amd
    net = _net(depth, start, 3, seed=7).train()
    before = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    x = torch.rand(*shape, dtype=torch.double, generator=torch.Generator().manual_seed(8))
    qiddm_amd.set_default_precision(precision)
    try:
        with torch.no_grad():
            got = net(x.cuda()).cpu()
    finally:
        qiddm_amd.set_default_precision("f32")
    want = ou.unet_simple_forward(x, before, depth, start, training=True)
    # batch statistics over 2 x 28 x 28 (or 4 x 10 x 10) values divide by small standard deviations: f32 looser
    tol = 2e-3 if precision == "f32" else 1e-8
    assert torch.allclose(got, want, atol=tol, rtol=tol), (got - want).abs().max()
    after = net.state_dict()
    moved = [k for k in before if k.endswith("running_mean") and not torch.equal(before[k], after[k].cpu())]
    assert len(moved) == 2 * depth - 1                         # every BatchNorm2d saw the batch
