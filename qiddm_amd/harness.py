"""Experiment harness: the build's counterpart of the reference drivers (src/mnist_exm.py and its
siblings).  Not a copy of those scripts -- they cannot run as checked in
(missing ``nn/__init__.py``, ``Log.py``, torchvision downloads) -- but it reproduces what a run depends on:

* seed order: ``torch.manual_seed`` / ``np.random.seed`` -> data -> ``first_x = rand(10,1,S,S)*0.75+0.5``
  -> model constructor draws (src/mnist_exm.py:369-371, 396, 424);
* model construction ``eval(f"nn.{name}")(*params)`` with digit strings cast to int (:420-424), the whole
  ``Diffusion`` cast to float64 (:443-449);
* ``DataLoader(TensorDataset(x_train), batch_size, shuffle=True)`` batch order (:404-408), Adam with the
  per-model learning rate (:170, :438), ``zero_grad -> diff(x, T, verbose=True) -> step`` (:179-182), the
  epoch loss as the sum of batch means (:178-185);
* checkpoint ``{'model_state_dict', 'loss_values', 'epochs'}`` at ``<save_path>/<save_name()>_<label>.pt``
  (:189-201) and resume ``start_epoch = checkpoint['epochs']`` (:294-323, 452-459);
* sampling: ``diff.sample(first_x, n_iters=15)``, clamp, x255 (:209-219).

Data: ``mnist_8x8`` (sklearn ``load_digits``, the one dataset of src/data.py reachable offline, :10-17) or
``synthetic_<S>`` uniform-noise images (what BASELINE.json prescribes for benchmarking).  With
``torchrun`` the training batch is sharded data-parallel and gradients are all-reduced
(``qiddm_amd.parallel``).

    python -m qiddm_amd.harness --model QNN_noise 64 4 2 --data mnist_8x8 --img_size 8 --batch_size 32 --epochs 2
"""
from __future__ import annotations

import argparse
import os
import pathlib
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import models, nn, noise, parallel

DEFAULT_LR = {  # src/mnist_exm.py:132-141
    "UNetUndirected": 0.01, "differN_noise": 0.00914, "QDenseUndirected_old_noise": 0.00211,
    "QIDDM_LL_noise": 0.0255, "QNN_noise": 0.01011, "QIDDM_PL_noise": 0.01116,
}


class Logger:
    """Tee for ``sys.stdout = Logger(path)`` (the reference imports it from a missing ``Log`` module,
    src/mnist_exm.py:19, 325-331)."""

    def __init__(self, path, stream=None):
        self.terminal = stream if stream is not None else sys.__stdout__
        self.log = open(path, "a")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        self.terminal.flush()
        self.log.flush()


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Quantum denoising diffusion on MI355X")
    p.add_argument("--model", nargs="+", default=["QNN_noise", "784", "8", "14"],
                   help="class name followed by its positional constructor arguments")
    p.add_argument("--data", type=str, default="synthetic_28", help="mnist_8x8 | synthetic_<side>")
    p.add_argument("--img_size", type=int, default=28)
    p.add_argument("--n_classes", type=int, default=10)
    p.add_argument("--label", type=int, default=0)
    p.add_argument("--reduced_size", type=float, default=1.0)
    p.add_argument("--load-path", type=str, default=None)
    p.add_argument("--save-path", type=str, default="results/run_")
    p.add_argument("--tau", type=int, default=10)
    p.add_argument("--target", type=str, default="data", choices=["data", "noise"])
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--ds-size", type=int, default=500)
    p.add_argument("--lr", type=float, default=None, help="default: the reference's per-model value")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--tau-test", type=int, default=15)
    p.add_argument("--log-dir", type=str, default=None)
    p.add_argument("--graph", action="store_true",
                   help="record the training step (fused step + one-launch Adam) into a HIP graph and replay it; "
                        "same numbers as the eager loop on the same noise")
    return p.parse_args(argv)


def mnist_8x8(n_classes=10, ds_size=100):
    from sklearn import datasets
    x, y = datasets.load_digits(n_class=n_classes, return_X_y=True)
    x = torch.tensor((x / 16).reshape(-1, 64), dtype=torch.double)
    y = torch.tensor(y, dtype=torch.long)
    return x[:ds_size], y[:ds_size], 8, 8


def synthetic(side, ds_size=100, label=0):
    x = torch.rand(ds_size, side * side, dtype=torch.double)
    return x, torch.full((ds_size,), label, dtype=torch.long), side, side


def load_data(args):
    if args.data == "mnist_8x8":
        return mnist_8x8(n_classes=args.n_classes, ds_size=args.ds_size)
    if args.data.startswith("synthetic_"):
        return synthetic(int(args.data.split("_")[1]), ds_size=args.ds_size, label=args.label)
    raise ValueError(f"unknown dataset {args.data!r} (offline: mnist_8x8, synthetic_<side>)")


def build_net(model_args):
    name = model_args[0]
    params = [int(a) if isinstance(a, str) and a.lstrip("-").isdigit() else a for a in model_args[1:]]
    return getattr(nn, name)(*params)


def load_model(diff, load_path, label):
    """Returns (loss_values, epochs_done); ([], 0) when there is no checkpoint (src/mnist_exm.py:294-323)."""
    lp = pathlib.Path(load_path) if str(load_path).endswith(".pt") else \
        pathlib.Path(load_path) / f"{diff.save_name()}_{label}.pt"
    try:
        checkpoint = torch.load(lp, map_location="cpu", weights_only=True)
    except FileNotFoundError:
        print("Failed to load model: File not found.\n")
        return [], 0
    diff.load_state_dict(checkpoint["model_state_dict"])
    print("Model loaded successfully.\n")
    return checkpoint["loss_values"], checkpoint["epochs"]


def train(diff, loader, args, start_epoch=0, loss_values=None):
    loss_values = list(loss_values or [])
    diff.train()
    use_graph = bool(getattr(args, "graph", False)) and str(args.device).startswith("cuda")
    if use_graph:
        from .optim import FusedAdam
        from .trainer import GraphedTrainStep
        opt = FusedAdam(diff.parameters(), lr=args.lr)
    else:
        opt = torch.optim.Adam(diff.parameters(), lr=args.lr)
    recorded = {}      # the shape of the first batch -> GraphedTrainStep; other shapes (an epoch's last, smaller
                       # batch) run the same fused step eagerly with the same optimizer
    world = dist.get_world_size() if dist.is_initialized() else 1
    dp_step = parallel.DataParallelStep(diff, opt) if world > 1 else None
    for _ in range(max(args.epochs - start_epoch, 0)):
        epoch_loss = torch.tensor(0.0, dtype=torch.double, device=args.device)
        for (batch,) in loader:
            x = batch.to(args.device, dtype=torch.double)
            if dp_step is not None:
                # every rank sees the global batch (same seeded loader), takes its contiguous shard -- possibly
                # uneven or empty -- and weights its gradient / loss by local_n / global_n
                lo, hi = parallel.shard_bounds(x.shape[0], dist.get_rank(), world)
                if use_graph and x.shape[0] >= world:
                    # --graph under data parallelism: the recorded step (gradient all-reduce inside the graph for RCCL)
                    # for the first global batch shape; the decision depends on the GLOBAL shape only, so every rank
                    # takes the same branch.  Other shapes (an epoch's last, smaller batch) run the eager DP step.
                    key = ("dp", tuple(x.shape))
                    if not recorded:
                        cpu_rng = torch.get_rng_state()
                        recorded[key] = GraphedTrainStep(diff, opt, x[lo:hi], T=args.tau, noise="reference",
                                                         shard=(x.shape[0], lo, hi))
                        torch.set_rng_state(cpu_rng)
                    step = recorded.get(key)
                    if step is not None:
                        epoch_loss += step(x[lo:hi])[0].mean() * ((hi - lo) / x.shape[0])
                        continue
                out = dp_step(x, T=args.tau, verbose=True)
                if out is not None:
                    epoch_loss += out[0].mean() * ((hi - lo) / x.shape[0])
                continue
            if use_graph:
                if not recorded:
                    cpu_rng = torch.get_rng_state()        # recording draws noise too: keep the stream of the run
                    recorded[tuple(x.shape)] = GraphedTrainStep(diff, opt, x, T=args.tau, noise="reference")
                    torch.set_rng_state(cpu_rng)
                step = recorded.get(tuple(x.shape))
                if step is not None:
                    epoch_loss += step(x)[0].mean()
                    continue
            opt.zero_grad()
            batch_loss, _ = diff(x=x, T=args.tau, verbose=True)
            epoch_loss += batch_loss.mean()
            opt.step()
        if dp_step is not None:
            red = epoch_loss if dist.get_backend() != "gloo" else epoch_loss.cpu()
            dist.all_reduce(red, op=dist.ReduceOp.SUM)      # sum of the weighted shard means = the global batch means
            epoch_loss = red
        loss_values.append(epoch_loss.item())
        print(f"epoch {len(loss_values)}: loss {loss_values[-1]:.6f}", flush=True)
    if args.epochs - start_epoch > 0 and (not dist.is_initialized() or dist.get_rank() == 0):
        sp = pathlib.Path(args.save_path) / f"{diff.save_name()}_{args.label}.pt"
        sp.parent.mkdir(parents=True, exist_ok=True)
        torch.save({"model_state_dict": diff.state_dict(), "loss_values": loss_values, "epochs": args.epochs}, sp)
    return loss_values


def test(diff, first_x, args):
    """Sampling + the reference's scaling to [0, 255] (src/mnist_exm.py:209-224):
    returns (tau_test + 1, batch, 1, H, W)."""
    diff.eval()
    mosaic = diff.sample(first_x=first_x, n_iters=args.tau_test, show_progress=False, only_last=False)
    mosaic = torch.clamp(torch.clamp(mosaic, 0.0, 1) * 255.0, 0.0, 255.0)
    it, s, b = args.tau_test + 1, args.img_size, first_x.shape[0]
    return mosaic.reshape(it, s, b, s).permute(0, 2, 1, 3).unsqueeze(2)


def main(argv=None):
    args = parse_args(argv)
    if args.log_dir:
        os.makedirs(args.log_dir, exist_ok=True)
        log = os.path.join(args.log_dir, "log-" + time.strftime("%Y%m%d-%H%M", time.localtime()) + ".log")
        sys.stdout = Logger(log)
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if args.device.startswith("cuda"):
            torch.cuda.set_device(local)
            args.device = f"cuda:{local}"
            dist.init_process_group("nccl", device_id=torch.device(args.device))
        else:
            dist.init_process_group("gloo")
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    x_all, y_all, height, width = load_data(args)
    if args.label is not None:
        x_all = x_all[y_all == args.label]
    x_all = x_all[: int(len(x_all) * args.reduced_size)].to(args.device, dtype=torch.double)
    cut = int(len(x_all) * 0.8)
    x_train, x_test = x_all[:cut], x_all[cut:]
    first_x = torch.rand(10, 1, args.img_size, args.img_size, dtype=torch.double).to(args.device) * 0.75 + 0.5
    args.batch_size = min(args.batch_size, max(len(x_train), 1))
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x_train.cpu()), batch_size=args.batch_size,
                                         shuffle=True)
    net = build_net(args.model)
    if args.lr is None:
        args.lr = DEFAULT_LR.get(args.model[0], 0.01)
    print(f"Initialized {args.model[0]} with parameters {args.model[1:]}, with {args.lr}")
    diff = models.Diffusion(net=net, noise_f=noise.add_normal_noise_multiple, prediction_goal=args.target,
                            shape=(height, width), loss=torch.nn.MSELoss()).to(args.device, dtype=torch.double)
    parallel.broadcast_parameters(diff)
    print("parameters:%d\n" % sum(p.numel() for p in diff.parameters() if p.requires_grad))
    loss_values, start_epoch = ([], 0) if args.load_path is None else load_model(diff, args.load_path, args.label)
    print(f"epoch start from {start_epoch}, left {args.epochs - start_epoch}")
    loss_values = train(diff, loader, args, start_epoch, loss_values)
    generated = test(diff, first_x, args)
    return diff, loss_values, generated, x_test


if __name__ == "__main__":
    main()
