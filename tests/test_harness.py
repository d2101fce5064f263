"""Harness: seed order, DataLoader order, checkpoint layout / file name,
resume.  The CPU part drives a classical net (qdepth = 0); the GPU part the C1 configurations
(MNIST 8x8, 4-qubit qdense, batch 32) of BASELINE.json on the one dataset reachable offline."""
import os

import pytest
import torch

from qiddm_amd import harness


def _argv(tmp, model, **kw):
    base = {"--data": "mnist_8x8", "--img_size": "8", "--batch_size": "4", "--epochs": "2", "--tau": "3",
            "--ds-size": "200", "--save-path": str(tmp), "--device": "cpu", "--tau-test": "4"}
    base.update({k: str(v) for k, v in kw.items()})
    out = ["--model"] + model
    for k, v in base.items():
        out += [k, v]
    return out


def test_offline_dataset_and_logger(tmp_path):
    x, y, h, w = harness.mnist_8x8(n_classes=10, ds_size=100)          # src/data.py:10-17
    assert x.shape == (100, 64) and x.dtype == torch.double and (h, w) == (8, 8)
    assert 0 <= x.min() and x.max() <= 1 and y.dtype == torch.long
    log = tmp_path / "l.log"
    lg = harness.Logger(str(log), stream=open(os.devnull, "w"))
    lg.write("hello\n")
    lg.flush()
    assert log.read_text() == "hello\n"
    with pytest.raises(ValueError):
        harness.load_data(harness.parse_args(["--data", "mnist_28x28"]))


def test_train_checkpoint_resume_cpu(tmp_path):
    model = ["UNetUndirected", "1", "4", "0"]
    diff, losses, gen, x_test = harness.main(_argv(tmp_path, model))
    ck_path = tmp_path / "unet_undirected_d1_s4_d0_0.pt"             # <save_name()>_<label>.pt
    assert ck_path.exists() and len(losses) == 2
    ck = torch.load(ck_path, weights_only=True)
    assert set(ck) == {"model_state_dict", "loss_values", "epochs"} and ck["epochs"] == 2
    assert all(k.startswith("net.") for k in ck["model_state_dict"])
    assert gen.shape == (5, 10, 1, 8, 8) and 0 <= gen.min() and gen.max() <= 255
    # same seed -> same trajectory (seed -> data -> first_x -> ctor draws -> shuffle order)
    _, losses2, gen2, _ = harness.main(_argv(tmp_path / "again", model))
    assert losses2 == losses and torch.equal(gen2, gen)
    # resume: epochs counts from the checkpoint
    _, losses3, _, _ = harness.main(_argv(tmp_path, model, **{"--epochs": 3, "--load-path": str(tmp_path)}))
    assert len(losses3) == 3 and losses3[:2] == losses
    assert torch.load(ck_path, weights_only=True)["epochs"] == 3
    # a finished run is not retrained (src/mnist_exm.py:172-173)
    _, losses4, _, _ = harness.main(_argv(tmp_path, model, **{"--epochs": 3, "--load-path": str(tmp_path)}))
    assert losses4 == losses3


@pytest.mark.gpu
@pytest.mark.parametrize("model,stem", [
    (["QNN_noise", "64", "4", "2"], "QNN_linear_features=4_qdepth=2_add_noise=0"),        # src/mnist_noise.py:49
    (["QIDDM_PL_noise", "64", "4", "2", "1"], "QIDDM_PL_noise=4_L=2_N=1"),                # src/mnist_noise.py:48
    (["differN_noise", "8", "4", "2"], "differN_old_pca=4_N=2_w8_h8_noise0"),              # src/mnist_noise.py:45
    (["QDenseUndirected_old_noise", "6", "8"], "QDenseUndirected_old_noise6_w8_h8_noise0"),
])
def test_c1_configurations_train_on_gpu(tmp_path, model, stem):
    """BASELINE config 1: MNIST 8x8, batch 32, the drivers' own model strings."""
    argv = _argv(tmp_path, model, **{"--device": "cuda", "--batch_size": 32, "--epochs": 3, "--tau": 10,
                                     "--target": "noise"})
    diff, losses, gen, _ = harness.main(argv)
    assert (tmp_path / f"{stem}_noise_0.pt").exists()                  # goal suffix, src/models.py:149-150
    assert len(losses) == 3 and all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] <= losses[0] * 1.05                              # Adam is not diverging
    assert gen.shape == (5, 10, 1, 8, 8)


@pytest.mark.gpu
def test_graph_mode_reproduces_the_eager_run(tmp_path):
    """--graph (recorded fused step + FusedAdam) == the eager loop: same loss curve, same trained weights."""
    common = ["--model", "QNN_noise", "64", "4", "2", "--data", "mnist_8x8", "--img_size", "8", "--batch_size", "16",
              "--epochs", "2", "--ds-size", "120", "--label", "0", "--tau", "5", "--device", "cuda"]
    d1, l1, g1, _ = harness.main(common + ["--save-path", str(tmp_path / "eager")])
    d2, l2, g2, _ = harness.main(common + ["--save-path", str(tmp_path / "graph"), "--graph"])
    assert l2 == pytest.approx(l1, rel=1e-6)
    for (k, a), (_, b) in zip(d1.state_dict().items(), d2.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-8), k
    assert torch.allclose(g1, g2, atol=1e-3)


@pytest.mark.gpu
def test_unet_simple_trains_through_the_harness_eager_and_recorded(tmp_path):
    """`UNetUndirectedS` by the drivers' model string (eval(f"nn.{name}")(*params), src/mnist_exm.py:424) on 8 x 8
    MNIST: the eager loop and the --graph loop (recorded step, fused Adam; QConv2d training through the circuit
    unitary, HIP BatchNorm) give the same loss curve and weights, and the checkpoint carries the reference's name."""
    common = ["--model", "UNetUndirectedS", "2", "4", "2", "--data", "mnist_8x8", "--img_size", "8", "--batch_size", "8",
              "--epochs", "2", "--ds-size", "48", "--label", "0", "--tau", "4", "--device", "cuda"]
    d1, l1, g1, _ = harness.main(common + ["--save-path", str(tmp_path / "eager")])
    d2, l2, g2, _ = harness.main(common + ["--save-path", str(tmp_path / "graph"), "--graph"])
    assert len(l1) == 2 and all(torch.isfinite(torch.tensor(l1)))
    assert l2 == pytest.approx(l1, rel=1e-4)
    for (k, a), (_, b) in zip(d1.state_dict().items(), d2.state_dict().items()):
        assert torch.allclose(a.double(), b.double(), rtol=1e-3, atol=1e-5), k
    assert any(p.name.startswith("unet_s_undirected_d2_s4_d2") for p in (tmp_path / "eager").parent.iterdir()) or \
        any("unet_s_undirected_d2_s4_d2" in p.name for p in tmp_path.rglob("*.pt"))
    assert g1.shape == g2.shape and torch.isfinite(g1).all()
