"""The lean sampling loop of the 8- and 6-qubit dense nets (qiddm_dense_sample_lean, csrc/qsim_lean.h) against the oracle:
tangent-form layers, fused DPP gates, the 8 x 8 composite of linear_down . linear_up for the steps after the first."""
import pytest
import torch

from oracle import circuits as oc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(N, L, S, P, seed, scale=0.6, n=8):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(7, P, generator=g, dtype=torch.float64)
    wd = torch.randn(n, P, generator=g, dtype=torch.float64) / P ** 0.5 * 3
    bd = torch.randn(n, generator=g, dtype=torch.float64)
    wu = torch.randn(P, n, generator=g, dtype=torch.float64) * 0.3
    bu = torch.rand(P, generator=g, dtype=torch.float64)
    w = torch.randn(N, L, S, n, 3, generator=g, dtype=torch.float64) * scale
    return x, wd, bd, wu, bu, w


def _oracle_steps(x, wd, bd, wu, bu, w, steps):
    spec = oc.Spec(n=w.shape[-2], encoding="rz", imprimitive="CZ", measure="expz")
    cur, refs = x, []
    for _ in range(steps):
        cur = oc.run_circuit(spec, cur @ wd.T + bd, w) @ wu.T + bu
        refs.append(cur)
    return torch.stack(refs)


# float32 bound: the general sampler's 1e-4 over three chained steps; the deepest case (2 x 27 simulated layers per step,
# weights of scale 0.6, every step feeding the next) sits at 2e-4 after FOUR steps in either kernel's arithmetic
@pytest.mark.parametrize("precision,tol", [("f32", 2.5e-4), ("f64", 1e-9)])
@pytest.mark.parametrize("n,N,L,S,P", [
    (8, 1, 1, 14, 784),     # the flagship QNN_noise(784, 8, 14): one block, angles are a global phase
    (8, 2, 6, 2, 784),      # QIDDM_LL-style, re-upload in front of blocks 1..5, two chained rounds
    (8, 1, 3, 3, 300),      # odd layer count per round (the two register sets end unevenly)
    (8, 2, 1, 1, 64),       # rounds without a simulated layer (product state -> read-out)
    (8, 1, 2, 1, 1500),     # every layer a block start; 8 pixels per thread
    (8, 3, 2, 2, 100),
    (6, 2, 14, 2, 784),     # QIDDM_LL_noise(784, 6, 14, 2): the reference's MNIST default (src/mnist_exm.py:46); no exchange
    (6, 1, 1, 14, 784),     # QNN_noise(784, 6, 14)
    (6, 2, 3, 3, 300),
    (6, 3, 1, 1, 64),
    (6, 1, 2, 2, 1500)])
def test_lean_sampler_matches_oracle(n, N, L, S, P, precision, tol):
    from qiddm_amd.circuit import Circuit, dense_sample, dense_sample_lean, dense_sample_lean_tables
    x, wd, bd, wu, bu, w = _case(N, L, S, P, seed=100 * N + 10 * L + S + n, n=n)
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=N, n_blocks=L, sel_layers=S)
    dev = lambda t: t.to(DEV)
    tables = dense_sample_lean_tables(circ, dev(w), dev(wd), dev(bd), dev(wu), dev(bu), precision)
    assert tables is not None, "weights of scale 0.6 are inside the tangent form's range"
    steps = 4
    got = dense_sample_lean(circ, dev(x), dev(wd), dev(bd), dev(wu), dev(bu), steps, tables, precision).cpu()
    ref = _oracle_steps(x, wd, bd, wu, bu, w, steps)
    assert got.shape == ref.shape
    assert torch.allclose(got, ref, atol=tol, rtol=tol), (got - ref).abs().max()
    # and against the general sampler on the same inputs (two independent kernels)
    gen = dense_sample(circ, dev(x), dev(wd), dev(bd), dev(w), dev(wu), dev(bu), steps, precision).cpu()
    assert torch.allclose(got, gen, atol=tol, rtol=tol), (got - gen).abs().max()


def _oracle_noise_steps(x, wd, bd, wu, bu, w, steps, nf):
    """The "noise"-goal loop body (reference src/models.py:130-134) around the oracle's net."""
    spec = oc.Spec(n=w.shape[-2], encoding="rz", imprimitive="CZ", measure="expz")
    cur, refs = x, []
    for _ in range(steps):
        net = oc.run_circuit(spec, cur @ wd.T + bd, w) @ wu.T + bu
        cur = torch.clamp(cur - (net - 0.5) * 0.1 * nf, 0, 1)
        refs.append(cur)
    return torch.stack(refs)


@pytest.mark.parametrize("precision,tol", [("f32", 5e-5), ("f64", 1e-10)])
@pytest.mark.parametrize("n,N,L,S,P", [
    (8, 1, 1, 14, 784), (8, 2, 6, 2, 784), (8, 1, 3, 3, 300), (8, 2, 1, 1, 64), (8, 1, 2, 1, 1500), (8, 3, 2, 2, 100),
    (6, 2, 14, 2, 784), (6, 1, 1, 14, 784), (6, 2, 3, 3, 300), (6, 3, 1, 1, 64), (6, 1, 2, 2, 1500)])
def test_lean_sampler_noise_goal_matches_oracle(n, N, L, S, P, precision, tol):
    """post_mode 1: x <- clamp(x - (net(x) - 0.5) * 0.1 * noise_factor, 0, 1), the image carried in registers; the update
    moves the image by a tenth of the net's output per step, hence the tighter float32 bound.  Six steps, so that a
    six-qubit net's alternating partial-sum buffers are each reused."""
    from qiddm_amd.circuit import Circuit, dense_sample, dense_sample_lean, dense_sample_lean_tables
    x, wd, bd, wu, bu, w = _case(N, L, S, P, seed=100 * N + 10 * L + S + n + 1, n=n)
    x = x * 1.2 - 0.1                                   # some pixels start outside [0, 1]; the clamp acts from step 1 on
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=N, n_blocks=L, sel_layers=S)
    dev = lambda t: t.to(DEV)
    tables = dense_sample_lean_tables(circ, dev(w), dev(wd), dev(bd), dev(wu), dev(bu), precision)
    assert tables is not None
    steps, nf = 6, 1.7
    got = dense_sample_lean(circ, dev(x), dev(wd), dev(bd), dev(wu), dev(bu), steps, tables, precision,
                            post_mode=1, noise_factor=nf).cpu()
    ref = _oracle_noise_steps(x, wd, bd, wu, bu, w, steps, nf)
    assert got.shape == ref.shape
    assert float(ref.min()) == 0.0 and float(ref.max()) == 1.0, "the case must exercise both clamp bounds"
    assert torch.allclose(got, ref, atol=tol, rtol=0), (got - ref).abs().max()
    gen = dense_sample(circ, dev(x), dev(wd), dev(bd), dev(w), dev(wu), dev(bu), steps, precision,
                       post_mode=1, noise_factor=nf).cpu()
    assert torch.allclose(got, gen, atol=tol, rtol=0), (got - gen).abs().max()


def test_lean_sampler_many_samples_and_strided_input():
    """More samples than workgroups in flight (grid-stride loop over samples) and an input with a row stride."""
    from qiddm_amd.circuit import Circuit, dense_sample_lean, dense_sample_lean_tables
    x, wd, bd, wu, bu, w = _case(2, 3, 2, 784, seed=5)
    g = torch.Generator().manual_seed(9)
    xb = torch.rand(2500, 800, generator=g, dtype=torch.float64)[:, :784]          # stride 800
    circ = Circuit(n_qubits=8, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=2, n_blocks=3, sel_layers=2)
    dev = lambda t: t.to(DEV)
    tables = dense_sample_lean_tables(circ, dev(w), dev(wd), dev(bd), dev(wu), dev(bu), "f32")
    xd = torch.empty(2500, 800, dtype=torch.float64, device=DEV)[:, :784]
    xd.copy_(xb)
    got = dense_sample_lean(circ, xd, dev(wd), dev(bd), dev(wu), dev(bu), 2, tables, "f32").cpu()
    idx = torch.tensor([0, 1, 255, 256, 1023, 2047, 2048, 2499])
    ref = _oracle_steps(xb[idx].contiguous(), wd, bd, wu, bu, w, 2)
    assert torch.allclose(got[:, idx], ref, atol=1e-4, rtol=1e-4), (got[:, idx] - ref).abs().max()


def test_weights_outside_the_tangent_range_fall_back_to_the_general_sampler():
    """theta / 2 within 3.5 degrees of 90: tan > 16 -> the tables are refused and the nets keep the general kernel; the
    result is still the oracle's."""
    import math
    from qiddm_amd import models, nn, noise
    from qiddm_amd.circuit import Circuit, dense_sample_lean_tables
    x, wd, bd, wu, bu, w = _case(1, 1, 4, 64, seed=3)
    w[0, 0, 2, 5, 1] = math.pi - 0.02                      # cos(theta / 2) = 0.01
    circ = Circuit(n_qubits=8, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=1, n_blocks=1, sel_layers=4)
    dev = lambda t: t.to(DEV)
    assert dense_sample_lean_tables(circ, dev(w), dev(wd), dev(bd), dev(wu), dev(bu), "f32") is None
    w_first = w.clone()
    w_first[0, 0, 2, 5, 1] = 0.3
    w_first[0, 0, 0, 5, 1] = math.pi - 0.02                # a round's FIRST layer is generated from (cos, sin): any angle
    assert dense_sample_lean_tables(circ, dev(w_first), dev(wd), dev(bd), dev(wu), dev(bu), "f32") is not None
    torch.manual_seed(0)
    net = nn.QNN_noise(64, 8, 4)
    with torch.no_grad():
        net.weights.copy_(w[0, 0])
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8)).to(DEV, dtype=torch.double).eval()
    img = torch.rand(5, 1, 8, 8, dtype=torch.float64)
    with torch.no_grad():
        got = diff.denoise_steps(img.to(DEV), 3).cpu()
    sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}
    ref = _oracle_steps(img.reshape(5, 64), sd["linear_down.weight"], sd["linear_down.bias"], sd["linear_up.weight"],
                        sd["linear_up.bias"], sd["weights"].reshape(1, 1, 4, 8, 3), 3)
    assert torch.allclose(got.reshape(3, 5, 64), ref, atol=1e-4), (got.reshape(3, 5, 64) - ref).abs().max()


@pytest.mark.parametrize("ctor", [lambda nn: nn.QNN_noise(784, 8, 14), lambda nn: nn.QIDDM_LL_noise(784, 8, 6, 2),
                                  lambda nn: nn.QIDDM_LL_noise(784, 6, 14, 2)])
def test_nets_route_goal_data_through_the_lean_kernel_and_track_weight_updates(ctor):
    """The nets' fused sampler uses the lean kernel for goal "data" (tables cached per weights AND linears); an in-place
    update of any of them rebuilds the tables; recording into a HIP graph after one eager call replays the same kernel."""
    from qiddm_amd import models, nn, noise
    torch.manual_seed(21)
    net = ctor(nn)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (28, 28)).to(DEV, dtype=torch.double).eval()
    x = (torch.rand(6, 1, 28, 28, dtype=torch.float64) * 0.75 + 0.5).to(DEV)

    def oracle():
        sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}
        w = sd["weights1"] if "weights1" in sd else sd["weights"].reshape((1, 1) + tuple(sd["weights"].shape))
        return _oracle_steps(x.cpu().reshape(6, 784), sd["linear_down.weight"], sd["linear_down.bias"],
                             sd["linear_up.weight"], sd["linear_up.bias"], w, 3).reshape(3, 6, 1, 28, 28)

    with torch.no_grad():
        got = diff.denoise_steps(x, 3)
        assert getattr(net, "_lean_tables_cache", (None, None))[1] is not None
        assert torch.allclose(got.cpu(), oracle(), atol=1e-4)
        net.linear_down.weight.mul_(1.5)                       # changes the composite map only
        net.linear_up.bias.add_(0.25)
        (net.weights1 if hasattr(net, "weights1") else net.weights).mul_(0.9)
        got2 = diff.denoise_steps(x, 3)
        assert torch.allclose(got2.cpu(), oracle(), atol=1e-4)
        assert not torch.allclose(got2, got, atol=1e-3)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            diff.denoise_steps(x, 3)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            rec = diff.denoise_steps(x, 3)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(rec, got2)


@pytest.mark.parametrize("ctor", [lambda nn: nn.QNN_noise(784, 8, 14), lambda nn: nn.QIDDM_LL_noise(784, 8, 6, 2),
                                  lambda nn: nn.QIDDM_LL_noise(784, 6, 14, 2)])
def test_nets_route_goal_noise_through_the_lean_kernel(ctor, monkeypatch):
    """Diffusion(..., "noise").denoise_steps: the same lean kernel with the clamp update; the general sampler must not run."""
    from qiddm_amd import circuit, models, nn, noise
    torch.manual_seed(22)
    net = ctor(nn)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "noise", (28, 28)).to(DEV, dtype=torch.double).eval()
    x = (torch.rand(5, 1, 28, 28, dtype=torch.float64) * 0.75 + 0.5).to(DEV)

    def refuse(*a, **k):
        raise AssertionError("the general sampler ran")
    monkeypatch.setattr(circuit, "dense_sample", refuse)
    with torch.no_grad():
        got = diff.denoise_steps(x, 4, noise_factor=0.8).cpu()
    sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}
    w = sd["weights1"] if "weights1" in sd else sd["weights"].reshape((1, 1) + tuple(sd["weights"].shape))
    ref = _oracle_noise_steps(x.cpu().reshape(5, 784), sd["linear_down.weight"], sd["linear_down.bias"],
                              sd["linear_up.weight"], sd["linear_up.bias"], w, 4, 0.8).reshape(4, 5, 1, 28, 28)
    assert torch.allclose(got, ref, atol=5e-5, rtol=0), (got - ref).abs().max()
