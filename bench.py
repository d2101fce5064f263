#!/usr/bin/env python
"""bench.py -- denoise-step throughput of the QIDDM quantum-layer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "denoise-step images/sec + gate-apps/sec, 8-qubit MNIST-28".
Workload at every N (weak scaling, one process per GPU, no data-path collective --
samples are independent): BASELINE configs[1] = MNIST 28x28, 8-qubit
qdense ``QNN_noise(784, 8, 14)`` (reference default model, src/mnist_exm.py:48), batch 256 per
GPU.  One step = one body of ``Diffusion.sample`` (reference src/models.py:127-134):
``x <- net(x)`` on a resident (256, 1, 28, 28) float64 batch, i.e.
linear_down -> [RZ encoders + 14 x (8 Rot + 8 CZ) + <Z>] -> linear_up; `--steps-per-graph` (15 = the
reference's n_iters per Diffusion.sample call, src/mnist_exm.py:211) consecutive steps of the sampling loop run in ONE launch of the fused sampler (qiddm_dense_sample:
four wavefronts per sample, the image stays in registers between steps and every intermediate image
is written out, as Diffusion.sample records it).  Synthetic
random-noise images (``rand*0.75+0.5``, src/mnist_exm.py:396), random-init weights under
``torch.manual_seed(42)``.  The step is captured once into a hipGraph and replayed.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
BATCH_PER_GPU = 256
N_QUBITS, QDEPTH, IMG = 8, 14, 28


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU per step")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--steps-per-graph", type=int, default=15,
                    help="consecutive denoise steps of the sampling loop per launch of the fused sampler "
                         "(15 = the reference's n_iters per Diffusion.sample call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary (non-headline) timings")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and the process "
                         "group is gloo (RCCL refuses two ranks on one device); timings are meaningless")
    return ap.parse_args()


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rehearse_on_one_gpu:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    return world, rank, local


def build_model(dev):
    from qiddm_amd import models, nn, noise
    torch.manual_seed(42)                                    # --seed default, src/mnist_exm.py:112
    net = nn.QNN_noise(IMG * IMG, N_QUBITS, QDEPTH)
    diff = models.Diffusion(net=net, noise_f=noise.add_normal_noise_multiple, prediction_goal="data",
                            shape=(IMG, IMG), loss=torch.nn.MSELoss()).to(dev, dtype=torch.double)
    return diff.eval()


def make_runner(diff, x0, use_graph, steps_per_graph, launches_per_graph=5):
    """Returns run(k): advance the resident batch by exactly k denoise steps.

    The sampling loop of the reference runs n_iters (=15, src/mnist_exm.py:211) dependent steps per call; one
    launch of the fused sampler holds `steps_per_graph` consecutive steps of that loop.  A recorded graph chains
    `launches_per_graph` such launches (each reads the previous launch's last image in place) and then refreshes
    the static input once; smaller graphs (one launch, one step) serve the remainder."""
    x = x0.clone()

    def chain(launches, m):
        with torch.no_grad():
            cur = x
            for _ in range(launches):
                cur = diff.denoise_steps(cur, m)[-1]   # m loop bodies (one launch when the net fuses them)
            x.copy_(cur)

    if not use_graph:
        def run_eager(k):
            for _ in range(k):
                chain(1, 1)
        return run_eager, x
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            chain(1, steps_per_graph)
    torch.cuda.current_stream().wait_stream(side)
    shapes = {(launches_per_graph, steps_per_graph), (1, steps_per_graph), (1, 1)}
    graphs = {}
    for launches, m in sorted(shapes):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain(launches, m)
        graphs[(launches, m)] = g

    def run(k):
        big, rest = divmod(k, launches_per_graph * steps_per_graph)
        mid, rest = divmod(rest, steps_per_graph)
        for _ in range(big):
            graphs[(launches_per_graph, steps_per_graph)].replay()
        for _ in range(mid):
            graphs[(1, steps_per_graph)].replay()
        for _ in range(rest):
            graphs[(1, 1)].replay()
    return run, x


def time_dominant_kernel(diff, x_dev, steps_per_launch, launches=100):
    """Average duration of the dominant kernel -- the fused sampler `dense_quad_kernel<float, 8, 4>`
    running `steps_per_launch` denoise steps per launch, exactly the launch of the timed region --
    measured with HIP events on the stream it is launched on: `launches` back-to-back launches
    inside one hipGraph replay."""
    circ = diff.net._circuit_descriptor()

    def once():
        with torch.no_grad():
            return diff.denoise_steps(x_dev, steps_per_launch)

    once()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        once()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(launches):
            once()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    avg_us = e0.elapsed_time(e1) * 1e3 / (reps * launches)
    return avg_us, circ


def _time_fn(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def secondary_measurements(dev, batch):
    """Other members of the same path on the same (batch, 1, 28, 28) shape -- reported next to the
    headline number, never instead of it: the re-uploading LL-style net, the
    unet_simple net, and training steps of the flagship (B*tau samples through fwd + bwd + Adam)."""
    from qiddm_amd import models, nn, noise
    out = {}
    x = (torch.rand(batch, 1, IMG, IMG, dtype=torch.double) * 0.75 + 0.5).to(dev)
    try:
        torch.manual_seed(42)
        ll = nn.QIDDM_LL_noise(IMG * IMG, 8, 6, 2).to(dev, dtype=torch.double).eval()
        # same measurement as the headline: the sampling loop, 10 consecutive steps per recorded launch
        ll_diff = models.Diffusion(ll, noise.add_normal_noise_multiple, "data", (IMG, IMG)).to(dev, dtype=torch.double).eval()
        run, _ = make_runner(ll_diff, x, True, 15)
        run(150)
        t = _time_fn(lambda: run(75), 20) / 75
        out["denoise_images_per_s_QIDDM_LL_noise(784,8,6,2)"] = batch / t
        out["gate_apps_per_s_QIDDM_LL_noise(784,8,6,2)"] = batch * 480 / t
    except Exception as e:  # pragma: no cover
        out["ll_error"] = repr(e)
    try:
        # the reference's own MNIST default (src/mnist_exm.py:46): 6 qubits, 14 x 2 layers, two rounds (G = 840)
        torch.manual_seed(42)
        ll6 = nn.QIDDM_LL_noise(IMG * IMG, 6, 14, 2).to(dev, dtype=torch.double).eval()
        ll6_diff = models.Diffusion(ll6, noise.add_normal_noise_multiple, "data", (IMG, IMG)).to(dev, dtype=torch.double).eval()
        run, _ = make_runner(ll6_diff, x, True, 15)
        run(150)
        t = _time_fn(lambda: run(75), 20) / 75
        out["denoise_images_per_s_QIDDM_LL_noise(784,6,14,2)"] = batch / t
        out["gate_apps_per_s_QIDDM_LL_noise(784,6,14,2)"] = batch * 840 / t
    except Exception as e:  # pragma: no cover
        out["ll6_error"] = repr(e)
    try:
        torch.manual_seed(42)
        unet = nn.UNetUndirectedS(3, 8, 3).to(dev, dtype=torch.double).eval()
        xb = x
        with torch.no_grad():
            t = _time_fn(lambda: unet(xb), 5, warm=1)
            try:    # the same forward recorded into a HIP graph (no allocations, no host work per replay)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    unet(xb)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    y_static = unet(xb)
                tg = _time_fn(g.replay, 20, warm=2)
                if torch.allclose(y_static, unet(xb), atol=1e-9):
                    out["denoise_images_per_s_UNetUndirectedS(3,8,3)_graphed"] = xb.shape[0] / tg
            except Exception as e:  # pragma: no cover
                out["unet_graph_error"] = repr(e)
        out["denoise_images_per_s_UNetUndirectedS(3,8,3)"] = xb.shape[0] / t
    except Exception as e:  # pragma: no cover
        out["unet_error"] = repr(e)
    try:
        # BASELINE config 4's layer: 12-qubit QConv2d(256 -> 256, 3x3, qdepth 3), eval mode = the one GEMM of the path
        # (implicit-im2col 65536 x 2304 by 2304 x 512 on the f32 MFMA); bound: mfma
        torch.manual_seed(42)
        conv = nn.QConv2d(256, 256, qdepth=3).to(dev).eval()
        xc = torch.rand(64, 256, 32, 32, dtype=torch.double, device=dev)
        with torch.no_grad():
            t = _time_fn(lambda: conv(xc), 5, warm=1)
        px = xc.shape[0] * 32 * 32
        tf = 2.0 * 2304 * 512 * px / t / 1e12
        out["qconv12_eval_pixels_per_s"] = px / t
        out["qconv12_roofline"] = {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                                   "kernel": "qiddm::qconv_gemm_wide_kernel"}
        del conv, xc
    except Exception as e:  # pragma: no cover
        out["qconv12_error"] = repr(e)
    for tag, detach in (("as_written_F1", True), ("parameter_shift", False), ("adjoint", False)):
        try:
            torch.manual_seed(42)
            net = nn.QNN_noise(IMG * IMG, N_QUBITS, QDEPTH, detach_quantum=detach)
            if tag == "adjoint":
                net.qnode.diff_method = "adjoint"     # the attribute the reference scripts poke
            if tag == "parameter_shift":
                net.fused_train_step = None           # the fused step always differentiates by the adjoint method:
                                                      # keep the reference's declared diff_method measurable
            diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (IMG, IMG),
                                    torch.nn.MSELoss()).to(dev, dtype=torch.double).train()
            opt = torch.optim.Adam(diff.parameters(), lr=1e-3)
            xt = x.reshape(batch, -1)[: (batch if tag != "parameter_shift" else min(batch, 32))]
            tau = 10

            def step():
                opt.zero_grad()
                diff(x=xt, T=tau)
                opt.step()
            t = _time_fn(step, 10 if tag != "parameter_shift" else 2, warm=1)
            out[f"train_images_per_s_{tag}"] = xt.shape[0] * tau / t
            if tag != "parameter_shift":
                # the same step (fused qiddm_train_step + one-launch Adam) recorded into a HIP graph, noise generated in the launch
                from qiddm_amd.optim import FusedAdam
                from qiddm_amd.trainer import GraphedTrainStep
                gstep = GraphedTrainStep(diff, FusedAdam(diff.parameters(), lr=1e-3), xt, T=tau, noise="fused")
                t = _time_fn(lambda: gstep(xt), 50, warm=3)
                out[f"train_images_per_s_{tag}_graphed"] = xt.shape[0] * tau / t
        except Exception as e:  # pragma: no cover
            out[f"train_error_{tag}"] = repr(e)
    try:
        # unet_simple training step (Diffusion loss, backward through the quantum convolutions, Adam) on a quarter of
        # the batch x tau = 10 noise levels: eager, then recorded into HIP graphs
        from qiddm_amd.optim import FusedAdam
        from qiddm_amd.trainer import GraphedTrainStep
        torch.manual_seed(42)
        unet_t = nn.UNetUndirectedS(3, 8, 3).to(dev, dtype=torch.double).train()
        diff = models.Diffusion(unet_t, noise.add_normal_noise_multiple, "data", (IMG, IMG),
                                torch.nn.MSELoss()).to(dev, dtype=torch.double).train()
        xt = x.reshape(batch, -1)[: max(batch // 4, 1)]
        opt = FusedAdam(diff.parameters(), lr=1e-3)

        def ustep():
            opt.zero_grad()
            diff(x=xt, T=10)
            opt.step()
        t = _time_fn(ustep, 5, warm=2)
        out["train_images_per_s_UNetUndirectedS(3,8,3)"] = xt.shape[0] * 10 / t
        gstep = GraphedTrainStep(diff, opt, xt, T=10, noise="device")
        t = _time_fn(lambda: gstep(xt), 20, warm=2)
        out["train_images_per_s_UNetUndirectedS(3,8,3)_graphed"] = xt.shape[0] * 10 / t
    except Exception as e:  # pragma: no cover
        out["unet_train_error"] = repr(e)
    return out


def cpu_baseline(diff, x0_cpu, budget_s):
    """The oracle (CPU restatement of default.qubit's per-gate complex128 update, batched) on the
    same denoise step, all host threads, bounded sample."""
    from oracle import circuits as oc
    sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}

    def net(t):
        return oc.qnn_forward(t, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights"],
                              sd["linear_up.weight"], sd["linear_up.bias"])

    x = x0_cpu
    with torch.no_grad():
        net(x)                                   # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            x = net(x)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 1000:
                break
    return n * x.shape[0] / el, n, el


def load_pmc_traffic(kernel_substr, batch):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        rec = json.load(open(path))
        for r in rec.get("kernels", []):
            if kernel_substr in r["kernel"] and r.get("batch") == batch:
                return r["hbm_bytes_per_launch"]
    except Exception:
        return None
    return None


def main():
    args = parse()
    world, rank, local = init_dist(args)
    dev = torch.device("cuda", local)
    from qiddm_amd import _capi
    _capi.lib()                                                  # fail loudly if the extension is missing

    diff = build_model(dev)
    torch.manual_seed(1000 + rank)
    x0 = (torch.rand(args.batch, 1, IMG, IMG, dtype=torch.double) * 0.75 + 0.5)
    run, _state = make_runner(diff, x0.to(dev), use_graph=not args.no_graph,
                              steps_per_graph=args.steps_per_graph)

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    images = world * args.batch * args.steps
    value = images / elapsed
    result = None
    if rank == 0:
        spl = 1 if args.no_graph else args.steps_per_graph
        kern_us, circ = time_dominant_kernel(diff, x0.to(dev), spl)
        g_per_sample = circ.gate_count()
        alg_bytes = circ.algorithmic_bytes_per_sample("f32") * args.batch * spl   # per launch
        achieved = alg_bytes / (kern_us * 1e-6) / 1e9
        result = {
            "metric": "denoise-step images/sec, 8-qubit MNIST-28",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "MNIST 28x28, 8-qubit qdense QNN_noise(784,8,14), batch 256 per GPU, "
                                   "one Diffusion.sample body (goal=data) per step",
                       "batch_per_gpu": args.batch, "global_batch": world * args.batch,
                       "n_qubits": N_QUBITS, "gates_per_sample": g_per_sample,
                       "launch": "eager" if args.no_graph else
                       f"hipGraph replay, {args.steps_per_graph} consecutive steps per launch, 5 launches per graph",
                       "parallelism": f"shard{world}"},
            "gate_apps_per_s": value * g_per_sample,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": load_pmc_traffic("dense_quad_kernel<float, 8, 4>", args.batch),
                         "kernel": "qiddm::dense_quad_kernel<float, 8, 4>",
                         "steps_per_launch": spl,
                         "kernel_avg_us": kern_us,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         # what the kernel actually executes (folded tables): per amplitude and layer one complex
                         # multiply (6 flop) + n real RY updates (6 flop each), plus <Z> and the two linears
                         "valu": (lambda fl: {"executed_flop_per_sample_step": fl,
                                              "achieved": fl * args.batch * spl / (kern_us * 1e-6) / 1e12,
                                              "peak": 157.3, "unit": "TFLOP/s",
                                              "frac": fl * args.batch * spl / (kern_us * 1e-6) / 1e12 / 157.3,
                                              "note": "latency-bound at batch 256: one sample per CU, one wavefront "
                                                      "per SIMD, every layer a dependent chain"})(
                             QDEPTH * (1 << N_QUBITS) * (6 + 6 * N_QUBITS) + 2 * N_QUBITS * (1 << N_QUBITS)
                             + 2 * 2 * IMG * IMG * N_QUBITS),
                         "note": "algorithmic = (G+1/2)*16*2^n B per sample (SURVEY 8d) x batch x steps per "
                                 "launch; the slab lives in registers (4 wavefronts per sample), so physical "
                                 "HBM traffic is the first image in + one image out per step; the kernel also "
                                 "does linear_down/linear_up of every step"},
        }
        if not args.no_secondary and world == 1:
            result["secondary"] = secondary_measurements(dev, args.batch)
        if not args.no_cpu_baseline and world == 1:
            v, n_steps, el = cpu_baseline(diff, x0, args.cpu_seconds)
            result["cpu_baseline"] = {
                "value": v, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{n_steps} denoise steps of the same batch-{args.batch} workload in {el:.1f} s "
                          "(oracle: batched complex128 per-gate torch update)",
                "gate_apps_per_s": v * g_per_sample}
            result["speedup_vs_cpu"] = value / v
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
