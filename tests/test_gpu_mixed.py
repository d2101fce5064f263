"""Density-matrix device (``qml.device("default.mixed")`` -> ``qiddm_mixed_forward``) against the oracle's dense
Kraus-operator simulation, on the circuits the reference's noise study runs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
NOISE = {1: None, 2: None, 3: None}


def _oracle_qnn_noise(x, weights, n, channel):
    from oracle import density as od
    rho = od.zero_rho(x.shape[0], n)
    for j in range(n):
        rho = od.rz_batched(rho, x[:, j], j, n)
        if channel is not None:
            rho = od.apply_kraus(rho, od.channel_kraus(*channel), j, n)
    rho = od.sel(rho, weights, n, "CZ")
    return od.expval_z(rho, n)


def _oracle_differn(x, weights, n, channel):
    from oracle import density as od
    rho = od.zero_rho(x.shape[0], n)
    for blk in range(weights.shape[0]):
        for j in range(n):
            rho = od.rz_batched(rho, x[:, j], j, n)
        rho = od.sel(rho, weights[blk], n, "CZ")
    if channel is not None:
        for j in range(n):
            rho = od.apply_kraus(rho, od.channel_kraus(*channel), j, n)
    return od.probs(rho)


@pytest.mark.parametrize("n", [1, 2, 4, 6, 7, 8])
@pytest.mark.parametrize("add_noise,channel", [(0, None), (1, ("PhaseDamping", 0.03)), (2, ("AmplitudeDamping", 0.05)),
                                               (3, ("DepolarizingChannel", 0.02))])
def test_qnn_noise_circuit_on_default_mixed(n, add_noise, channel):
    from qiddm_amd import nn, qml
    torch.manual_seed(n * 10 + add_noise)
    net = nn.QNN_noise(16, n, 2, add_noise=add_noise).to(DEV)
    # what src/mnist_noise.py:214-229 does to the layer before sampling
    net.qdev = qml.device("default.mixed", wires=n)
    net.qnode = qml.QNode(net._circuit, net.qdev, interface="torch", diff_method="backprop")
    x = torch.randn(5, n, dtype=torch.float64, device=DEV)
    want = _oracle_qnn_noise(x.cpu(), net.weights.detach().cpu(), n, channel)
    for prec, tol in (("f64", 1e-11), ("f32", 3e-5)):
        net.qnode.precision = prec
        got = net.qnode(x, net.weights)
        assert got.dtype == torch.float64 and got.shape == (5, n)
        assert (got.cpu() - want).abs().max().item() < tol, prec
    one = net.qnode(x[0], net.weights)                                  # per-sample call, as the reference loops
    assert one.shape == (n,) and (one.cpu() - want[0]).abs().max().item() < 3e-5


@pytest.mark.parametrize("n,channel", [(3, None), (5, ("AmplitudeDamping", 0.1)), (6, ("DepolarizingChannel", 0.02)),
                                       (8, ("AmplitudeDamping", 0.1))])
def test_differn_style_circuit_with_trailing_channels(n, channel):
    from qiddm_amd import qml
    torch.manual_seed(n)
    w = (torch.randn(2, 2, n, 3, dtype=torch.float64) * 0.5).to(DEV)
    x = torch.randn(4, n, dtype=torch.float64, device=DEV)
    dev = qml.device("default.mixed", wires=n)

    def circuit(inputs, weights):
        for i in range(2):
            for j in range(n):
                qml.RZ(inputs[:, j], wires=j)
            qml.StronglyEntanglingLayers(weights[i], wires=range(n), imprimitive=qml.ops.CZ)
        if channel is not None:
            for j in range(n):
                getattr(qml, channel[0])(channel[1], wires=j)
        return qml.probs(wires=range(n))

    got = qml.QNode(circuit, dev, interface="torch", diff_method="backprop", precision="f64")(x, w)
    want = _oracle_differn(x.cpu(), w.cpu(), n, channel)
    assert (got.cpu() - want).abs().max().item() < 1e-11
    assert torch.allclose(got.sum(dim=1), torch.ones(4, dtype=torch.float64, device=DEV), atol=1e-12)   # trace 1


def test_amplitude_embedding_cnot_circuit_matches_pure_state_without_noise():
    """QDenseUndirected_old_noise's circuit (amplitude embedding, CNOT rings): with add_noise = 0 the density-matrix
    device must reproduce the statevector kernel; with AmplitudeDamping(0.1) the oracle's Kraus sum."""
    from oracle import density as od
    from oracle import statevector as sv
    from qiddm_amd import nn, qml
    torch.manual_seed(2)
    for add_noise in (0, 2):
        net = nn.QDenseUndirected_old_noise(3, 4, add_noise=add_noise).to(DEV).double()  # 4x4 image -> n = 4
        x = torch.rand(3, 16, dtype=torch.float64, device=DEV)
        pure = net.qnode(x) if add_noise == 0 else None
        net.qdev = qml.device("default.mixed", wires=net.wires)
        net.qnode = qml.QNode(net._circuit, net.qdev, interface="torch", diff_method="backprop", precision="f64")
        got = net.qnode(x)
        psi = sv.amplitude_embedding(x.cpu(), 4, pad_with=0.1, normalize=True)
        rho = od.sel(od.from_state(psi, 4), torch.tanh(net.weights.detach().cpu()), 4, "CNOT")
        if add_noise == 2:
            for j in range(4):
                rho = od.apply_kraus(rho, od.channel_kraus("AmplitudeDamping", 0.1), j, 4)
        assert (got.cpu() - od.probs(rho)).abs().max().item() < 1e-11
        if pure is not None:
            assert (got - pure).abs().max().item() < 2e-5                                # f32 statevector kernel


def test_known_answers_and_errors():
    from qiddm_amd import qml
    dev = qml.device("default.mixed", wires=1)
    one = torch.ones(1, dtype=torch.float64, device=DEV)

    def flip_then(channel, p):
        def circuit(t):
            qml.RY(t * math.pi, wires=0)                       # |1>
            getattr(qml, channel)(p, wires=0)
            return qml.probs(wires=range(1))
        return qml.QNode(circuit, dev, interface="torch", precision="f64")(one)[0].cpu()

    assert torch.allclose(flip_then("AmplitudeDamping", 1.0), torch.tensor([1.0, 0.0], dtype=torch.float64), atol=1e-12)
    assert torch.allclose(flip_then("DepolarizingChannel", 0.75), torch.tensor([0.5, 0.5], dtype=torch.float64), atol=1e-12)
    assert torch.allclose(flip_then("PhaseDamping", 0.7), torch.tensor([0.0, 1.0], dtype=torch.float64), atol=1e-12)
    # a pure-state device still refuses channels, as PennyLane does
    pure = qml.device("default.qubit.torch", wires=1)

    def noisy(t):
        qml.RY(t, wires=0)
        qml.PhaseDamping(0.1, wires=0)
        return qml.probs(wires=range(1))
    with pytest.raises(qml.DeviceError):
        qml.QNode(noisy, pure, interface="torch")(one)
    big = qml.device("default.mixed", wires=9)

    def nine(t):
        qml.RY(t, wires=0)
        return qml.probs(wires=range(9))
    from qiddm_amd._capi import QiddmError
    with pytest.raises(QiddmError):
        qml.QNode(nine, big, interface="torch")(one)
    with pytest.raises(RuntimeError):
        qml.QNode(noisy, dev, interface="torch")(torch.ones(1, dtype=torch.float64))     # CPU tensor: no CPU path


def test_noise_study_flow_end_to_end():
    """src/mnist_noise.py:214-229: re-bind the trained layer to default.mixed, switch the channel on, sample."""
    from oracle import diffusion as odf
    from qiddm_amd import models, nn, noise, qml
    torch.manual_seed(4)
    net = nn.QNN_noise(64, 4, 2)
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (8, 8)).to(DEV, dtype=torch.double).eval()
    first_x = (torch.rand(3, 1, 8, 8, dtype=torch.double) * 0.75 + 0.5).to(DEV)
    clean = diff.sample(first_x=first_x, n_iters=2, only_last=True)
    diff.net.device_type, diff.net.diff_method = "default.mixed", "backprop"
    diff.net.add_noise = 3
    diff.net.qdev = qml.device(diff.net.device_type, wires=diff.net.hidden_features)
    diff.net.qnode = qml.QNode(diff.net._circuit, diff.net.qdev, interface="torch", diff_method=diff.net.diff_method)
    noisy = diff.sample(first_x=first_x, n_iters=2, only_last=True)
    sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}

    def ref_net(t):
        xr = t.reshape(t.shape[0], -1) @ sd["linear_down.weight"].T + sd["linear_down.bias"]
        ev = _oracle_qnn_noise(xr, sd["weights"], 4, ("DepolarizingChannel", 0.02))
        return (ev @ sd["linear_up.weight"].T + sd["linear_up.bias"]).reshape(t.shape)

    want = odf.denoise_step(ref_net, odf.denoise_step(ref_net, first_x.cpu()))
    assert (noisy.cpu() - want).abs().max().item() < 1e-4
    assert not torch.allclose(noisy, clean, atol=1e-4)            # the channel changed the images
