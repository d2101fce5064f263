#!/usr/bin/env python
"""bench.py -- denoise-step throughput of the QIDDM quantum-layer hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): "denoise-step images/sec + gate-apps/sec, 8-qubit MNIST-28".
Workload at every N (weak scaling, one process per GPU, no data-path collective in the denoise step --
samples are independent): BASELINE configs[1] = MNIST 28x28, 8-qubit
qdense ``QNN_noise(784, 8, 14)`` (reference default model, src/mnist_exm.py:48), batch 256 per
GPU.  One step = one body of ``Diffusion.sample`` (reference src/models.py:127-134):
``x <- net(x)`` on a resident (256, 1, 28, 28) float64 batch, i.e.
linear_down -> [RZ encoders + 14 x (8 Rot + 8 CZ) + <Z>] -> linear_up.  Consecutive steps of the sampling loop run
inside ONE launch of the fused sampler (qiddm_dense_sample: four wavefronts per sample, the image stays in
registers between steps and every intermediate image is written out, as Diffusion.sample records it):
15 per launch (the reference's n_iters per Diffusion.sample call, src/mnist_exm.py:211), a remainder of K rides in
the last launch.  Synthetic random-noise images (``rand*0.75+0.5``, src/mnist_exm.py:396), random-init weights
under ``torch.manual_seed(42)``.  The K steps are recorded into hipGraphs and replayed; the timed region repeats
the K steps until it is >= 50 ms long (``repeats``), bracketed by barrier + synchronize, max over ranks.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = 256


def parse(argv=None):
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
    ap.add_argument("--no-train", action="store_true", help="skip the data-parallel training-step timing")
    ap.add_argument("--no-f64", action="store_true", help="skip the second timed region in the reference's precision")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget per variant")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses cuda:0 and the process "
                         "group is gloo (RCCL refuses two ranks on one device); timings are meaningless")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="print the command `--gpus N` would start (JSON) and exit; touches no GPU")
    return ap.parse_args(argv)


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(args, argv, port=None):
    """The N-rank command `python bench.py --gpus N ...` stands for when no launcher started it: one process per GPU
    under torch.distributed.run on this node, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.join(ROOT, "bench.py")] + [a for a in argv if a != "--dry-run-launch"]


def self_launch_if_needed(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks as a CHILD process (this
    parent has not touched the GPU -- torch is not even imported yet -- and never does), pass its output through and
    leave with its return code."""
    args = parse(argv)
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.dry_run_launch:
        print(json.dumps({"self_launch": args.gpus > 1 and not launched,
                          "command": launch_command(args, argv, port=29500) if args.gpus > 1 else None}))
        sys.exit(0)
    if args.gpus > 1 and not launched:
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        sys.exit(subprocess.call(launch_command(args, argv), env=env))


if __name__ == "__main__":
    self_launch_if_needed(sys.argv[1:])

import torch  # noqa: E402  (after the self-launch decision: the parent of an N-rank run never loads it)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_TF = 157.3           # f32 vector peak (= the f32 MFMA peak), same guide
N_QUBITS, QDEPTH, IMG = 8, 14, 28
MIN_TIMED_S = 0.05
HEADLINE_KERNEL = "qiddm::dense_lean_kernel<{}, 8, 4, false, 14, false>"
TRAFFIC_PROFILE = "profiles/r03c/bench_pmc_traffic.json"     # FETCH_SIZE / WRITE_SIZE passes of the driver's command
HEADLINE_NOTE = ("latency-bound at batch 256: one sample per CU, one wavefront per SIMD, and a layer is ONE dependent chain "
                 "(every gate acts on the same 256 amplitudes): ~11 cycles per dependent vector instruction for a lone "
                 "wavefront + ~230 for the LDS exchange of the two wave-bit gates (tools/ubench/). All 14 layers of every "
                 "step are computed (the first acts on |0..0>: a real product state, tabulated per weights); RY gates run in "
                 "tangent form, one fused cross-lane multiply-add each (4 flop per amplitude and gate instead of 6: "
                 "`executed_flop`; `standard_form_flop` is the round-2 count of the same circuit). linear_down is not "
                 "evaluated: its output enters this circuit only as RZ on |0..0>, a global phase (finding F2) -- for nets that "
                 "re-upload it is composed with linear_up into an n x n map of the previous step's <Z>")


def lean_flop(n, layers_per_round, rounds, pixels, reupload):
    """Executed flop per sample and denoise step of the lean sampler (qsim_lean.h): per amplitude and SIMULATED layer one
    complex multiply (6) + n tangent-form RYs (one multiply-add on each of re / im: 4); per round the read-out (|a|^2: 3,
    n signed sums: n); linear_up (2 P n); the n x n composite of the next step's angles where the circuit re-uploads."""
    d = 1 << n
    return rounds * ((layers_per_round - 1) * d * (6 + 4 * n) + d * (3 + n)) + 2 * pixels * n + (2 * n * n if reupload else 0)


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: the line "
                         f"would report the wrong n_gpus; start `python bench.py --gpus {args.gpus}` (it launches its "
                         f"own ranks) or torch.distributed.run with --nproc-per-node {args.gpus}")
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
            assert dist.get_backend() == "nccl", dist.get_backend()     # "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def _max_over_ranks(v: float, dev) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return v
    t = torch.tensor([v], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def build_model(dev):
    from qiddm_amd import models, nn, noise
    torch.manual_seed(42)                                    # --seed default, src/mnist_exm.py:112
    net = nn.QNN_noise(IMG * IMG, N_QUBITS, QDEPTH)
    diff = models.Diffusion(net=net, noise_f=noise.add_normal_noise_multiple, prediction_goal="data",
                            shape=(IMG, IMG), loss=torch.nn.MSELoss()).to(dev, dtype=torch.double)
    return diff.eval()


def launch_plan(k, spl):
    """Steps per launch for exactly k consecutive steps: launches of `spl`, the remainder rides in the last one."""
    if k <= 0:
        return []
    full, rest = divmod(k, spl)
    if full == 0:
        return [rest]
    plan = [spl] * full
    plan[-1] += rest
    return plan


def hip_graph_recorder(fn):
    """Run `fn` once on a side stream (lazily sized buffers of an unseen launch shape), then record it into a hipGraph."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


class Runner:
    """run(k): advance the resident batch by exactly k denoise steps.

    A recorded graph holds up to `launches_per_graph` launches of the fused sampler (each reads the previous launch's
    last image in place) and then refreshes the static input once.  Graphs are recorded per distinct launch plan --
    ahead of the timed region by `prepare`; `run` records a plan it has not seen (so a caller that prepares one step
    count and runs another still gets exactly k steps, only the first such call pays the recording).
    `recorder(fn)` returns an object with `.replay()`; the default records a hipGraph (tests inject a fake)."""

    GROUP = 8   # k-step blocks per recorded graph of run_repeated

    def __init__(self, diff, x0, use_graph, spl, launches_per_graph=8, recorder=None):
        self.diff, self.x, self.use_graph, self.spl, self.lpg = diff, x0.clone(), use_graph, spl, launches_per_graph
        self.recorder = recorder or hip_graph_recorder
        self.graphs = {}
        self.steps_done = 0          # bookkeeping the tests (and the line's sanity check) read
        self.lazy_records = 0

    def _chain(self, plan):
        with torch.no_grad():
            cur = self.x
            for m in plan:
                cur = self.diff.denoise_steps(cur, m)[-1]   # m loop bodies (one launch when the net fuses them)
            self.x.copy_(cur)

    def _chunks(self, k):
        plan = launch_plan(k, self.spl)
        return [tuple(plan[i:i + self.lpg]) for i in range(0, len(plan), self.lpg)]

    def _graph(self, chunk, lazy=False):
        g = self.graphs.get(chunk)
        if g is None:
            g = self.graphs[chunk] = self.recorder(lambda: self._chain(chunk))
            self.lazy_records += bool(lazy)
        return g

    def prepare(self, k):
        if self.use_graph:
            for chunk in set(self._chunks(k)):
                self._graph(chunk)

    def run(self, k):
        if not self.use_graph:
            for _ in range(k):
                self._chain((1,))
        else:
            for chunk in self._chunks(k):
                self._graph(chunk, lazy=True).replay()
        self.steps_done += max(k, 0)

    def prepare_repeated(self, k):
        """When k steps are ONE launch, a graph of GROUP consecutive k-step launches (each continuing from the previous
        one's last image, the static input refreshed once at the end) serves run_repeated."""
        plan = tuple(launch_plan(k, self.spl))
        if self.use_graph and len(plan) == 1:
            self._graph(plan * self.GROUP)

    def run_repeated(self, k, reps):
        """Exactly reps * k denoise steps: groups of GROUP k-step launches per graph replay where one was recorded
        (inside a graph the launches follow each other without a host round trip), single k-step replays for the rest."""
        key = tuple(launch_plan(k, self.spl)) * self.GROUP
        g = self.graphs.get(key) if self.use_graph and k > 0 else None
        full, rest = divmod(reps, self.GROUP) if g is not None else (0, reps)
        for _ in range(full):
            g.replay()
        self.steps_done += full * self.GROUP * max(k, 0)
        for _ in range(rest):
            self.run(k)


def timed_region(runner, steps, warmup, world, dev):
    """W untimed steps, then `repeats` x exactly K steps between barrier + synchronize; returns (seconds, repeats)."""
    runner.prepare(steps)
    runner.prepare(warmup)
    runner.prepare_repeated(steps)
    runner.run(warmup)
    torch.cuda.synchronize()
    # calibrate the repeat count on untimed passes of the K steps (the first replay of a fresh graph is slow: time the
    # third); every rank must use the same count
    for _ in range(2):
        runner.run(steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    runner.run(steps)
    torch.cuda.synchronize()
    once = _max_over_ranks(time.perf_counter() - t0, dev)
    repeats = max(1, int(math.ceil(MIN_TIMED_S / max(once, 1e-7))))
    while True:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run_repeated(steps, repeats)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = _max_over_ranks(time.perf_counter() - t0, dev)
        # back-to-back replays run faster than the calibration pass: if the region came out short, time a longer one
        # (every rank sees the same max-over-ranks figure, so all of them repeat together)
        if elapsed >= MIN_TIMED_S or repeats >= (1 << 20):
            return elapsed, repeats
        repeats = int(math.ceil(repeats * 1.5 * MIN_TIMED_S / max(elapsed, 1e-7)))


def time_dominant_kernel(diff, x_dev, steps_per_launch, launches=100):
    """Average duration of the dominant kernel -- the fused sampler `dense_quad_kernel<T, 8, 4>`
    running `steps_per_launch` denoise steps per launch, exactly the launch of the timed region --
    measured with HIP events on the stream it is launched on: `launches` back-to-back launches
    inside one hipGraph replay."""
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
    return e0.elapsed_time(e1) * 1e3 / (reps * launches)


def _time_fn(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def _event_time_us(fn, iters, warm=2):
    """HIP events on torch's current stream -- the stream every qiddm launch of `fn` goes to."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def dense_flop(n, layers_per_round, rounds, pixels):
    """Executed flop per sample and denoise step of a linear_down -> circuit -> linear_up net on the folded tables:
    per amplitude and SIMULATED layer one complex multiply (6 flop) + n real RY updates (6 flop each); the first layer of
    a round acts on |0..0> and is generated as a product state (n multiplies per amplitude); <Z>; the two linears."""
    d = 1 << n
    return (rounds * ((layers_per_round - 1) * d * (6 + 6 * n) + n * d + 2 * n * d) + 2 * 2 * pixels * n)


def f64_timed(dev, x0, args, spl, world):
    """The SAME timed region a second time in the reference's own precision (complex128 statevector + float64 linears,
    finding F5; src/mnist_exm.py:449): W warm-up steps, `repeats` x K steps between barrier + synchronize, max over
    ranks -- plus the kernel's own duration from HIP events."""
    import qiddm_amd
    qiddm_amd.set_default_precision("f64")
    try:
        diff = build_model(dev)
        runner = Runner(diff, x0.to(dev), use_graph=not args.no_graph, spl=spl)
        elapsed, repeats = timed_region(runner, args.steps, args.warmup, world, dev)
        plan = launch_plan(args.steps, spl)
        kspl = max(set(plan), key=plan.count) if plan else spl
        us = time_dominant_kernel(diff, x0.to(dev), kspl, launches=20)
    finally:
        qiddm_amd.set_default_precision("f32")
    total = args.steps * repeats
    per_step = elapsed / total
    images = world * x0.shape[0] / per_step
    flop = lean_flop(N_QUBITS, QDEPTH, 1, IMG * IMG, False)
    tf = flop * x0.shape[0] * kspl / (us * 1e-6) / 1e12
    return {"dtype": "f64", "kernel": HEADLINE_KERNEL.format("double"), "steps_per_launch": kspl,
            "ms_per_step": per_step * 1e3, "timed_region_s": elapsed, "repeats": repeats,
            "images_per_s": images, "gate_apps_per_s": images * 232,
            "kernel_avg_us": us, "kernel_us_per_step": us / kspl,
            "roofline": {"bound": "valu", "achieved": tf, "peak": VALU_PEAK_TF / 2, "unit": "TFLOP/s",
                         "frac": tf / (VALU_PEAK_TF / 2), "note": "f64 vector peak = half the f32 one (78.6 TFLOP/s)"},
            "note": "complex128 statevector + float64 linears: the reference's own precision (src/mnist_exm.py:449); "
                    "images_per_s is from the timed region (same bracket as `value`), kernel_* from HIP events"}


def c5_roofline(dev):
    """BASELINE configs[4]'s per-GPU shard -- the one configuration whose statevector does not fit on chip, so slab
    sweeps are physical memory traffic: 16-qubit LL-style circuit of ``(2352, 16, 6, 2)`` (two chained rounds of six
    [RZ(x) ; SEL(2 layers, CZ)] blocks, <Z>), 1024 samples, kernel `wide_cz_kernel` / `tiled_circuit_kernel`.
    `achieved` = the kernel's own sweep bytes (one read + one write of the 2^n-amplitude slab per pass) / launch
    time; the in-profile FETCH/WRITE counters (profiles/) say how much of that reaches the memory side."""
    from qiddm_amd.circuit import Circuit, prepare_gates, run_forward, wide_sweeps_per_sample
    torch.manual_seed(3)
    n, batch = 16, 1024
    circ = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=2, n_blocks=6, sel_layers=2)
    w = (torch.randn(circ.angles_shape, dtype=torch.float64) * 0.4).to(dev)
    x = torch.randn(batch, n, device=dev)
    table = prepare_gates(circ, w, "f32")
    us = _event_time_us(lambda: run_forward(circ, x, w, "f32", table=table), iters=5, warm=2)
    # the reverse sweep of ONE round of the same shard (weights + input gradients; wide_cz_adjoint_kernel)
    try:
        from qiddm_amd.circuit import run_adjoint
        one = Circuit(n_qubits=n, encoding="rz", imprimitive="CZ", measure="expz", n_rounds=1, n_blocks=6, sel_layers=2)
        g = torch.randn(batch, n, device=dev)
        adj_us = _event_time_us(lambda: run_adjoint(one, x, w[0:1], g, "f32"), iters=3, warm=1)
    except Exception as e:  # pragma: no cover
        adj_us = repr(e)
    sweeps, kernel = wide_sweeps_per_sample(circ, "f32")
    sweep_bytes = sweeps * 2 * 8 * (1 << n) * batch                    # read + write of every complex64 amplitude
    alg = circ.algorithmic_bytes_per_sample("f32") * batch
    ach = sweep_bytes / (us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": None, "kernel": kernel, "workload": "C5 shard: 16-qubit LL-style (2352,16,6,2) circuit, 1024 samples",
            "kernel_avg_us": us, "circuits_per_s": batch / (us * 1e-6),
            "gate_apps_per_s": batch * circ.gate_count() / (us * 1e-6),
            "sweeps_per_sample": sweeps, "sweep_bytes_per_launch": sweep_bytes,
            "adjoint_one_round_us": adj_us,
            "hbm_equivalent": {"bytes_per_launch": alg, "GBps": alg / (us * 1e-6) / 1e9,
                               "note": "SURVEY 8d per-gate model (16*2^n B per gate application); exceeds the HBM peak "
                                       "because one sweep applies a whole layer"},
            "note": "traffic: see profiles/ (separate --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_wide.py)"}


def _graph_event_us(fn, launches=20, reps=3):
    """Average duration of `fn` (one or more launches on torch's current stream): `launches` copies recorded into one
    hipGraph, HIP events around `reps` replays."""
    g = hip_graph_recorder(lambda: [fn() for _ in range(launches)])
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * launches)


def _valu_block(flop_per_launch, us, kernel, note=""):
    tf = flop_per_launch / (us * 1e-6) / 1e12
    return {"bound": "valu", "achieved": tf, "peak": VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TF,
            "kernel": kernel, "kernel_avg_us": us, **({"note": note} if note else {})}


def secondary_dense_samplers(dev, out):
    """The other dense nets of the path through the same fused sampling loop (reference drivers' own parameter lists),
    each with a small roofline block for its launch (15 steps per launch; HIP events)."""
    from qiddm_amd import models, nn, noise
    down = 2 * IMG * IMG * 8          # the "noise" goal runs the whole linear_down every step (no composite map)
    cases = (
        # tag, ctor, batch, image side, gates/sample, kernel, executed flop per sample-step, prediction goal
        ("QIDDM_LL_noise(784,8,6,2)", lambda: nn.QIDDM_LL_noise(IMG * IMG, 8, 6, 2), 256, IMG, 480,          # src/fashion_exm.py:45 (LL form)
         "qiddm::dense_lean_kernel<float, 8, 4, true, 12, false>", lean_flop(8, 12, 2, IMG * IMG, True), "data"),
        ("QIDDM_LL_noise(784,6,14,2)", lambda: nn.QIDDM_LL_noise(IMG * IMG, 6, 14, 2), 256, IMG, 840,        # src/mnist_exm.py:46
         "qiddm::dense_lean_kernel<float, 6, 4, true, 28, false>", lean_flop(6, 28, 2, IMG * IMG, True), "data"),
        ("QIDDM_LL_noise(784,8,6,2)_goal_noise", lambda: nn.QIDDM_LL_noise(IMG * IMG, 8, 6, 2), 256, IMG, 480,
         "qiddm::dense_lean_kernel<float, 8, 4, true, 12, true>", lean_flop(8, 12, 2, IMG * IMG, True) - 128 + down,
         "noise"),                                                                                          # src/models.py:130-134
        ("C1_QNN_noise(64,4,2)_b32", lambda: nn.QNN_noise(64, 4, 2), 32, 8, 20,                              # src/mnist_noise.py:49
         "qiddm::dense_quad_kernel<float, 4, 4>", dense_flop(4, 2, 1, 64), "data"),
    )
    for tag, ctor, batch, side, gates, kernel, flop, goal in cases:
        try:
            torch.manual_seed(42)
            net = ctor().to(dev, dtype=torch.double).eval()
            d = models.Diffusion(net, noise.add_normal_noise_multiple, goal, (side, side)).to(dev, dtype=torch.double).eval()
            x = (torch.rand(batch, 1, side, side, dtype=torch.double) * 0.75 + 0.5).to(dev)
            r = Runner(d, x, True, 15)
            r.prepare(75)
            r.run(150)
            t = _time_fn(lambda: r.run(75), 20) / 75
            out[f"denoise_images_per_s_{tag}"] = batch / t
            out[f"gate_apps_per_s_{tag}"] = batch * gates / t
            with torch.no_grad():
                us = _graph_event_us(lambda: d.denoise_steps(x, 15))
            out[f"roofline_{tag}"] = _valu_block(flop * batch * 15, us, kernel,
                                                 "15 denoise steps per launch; latency-bound like the headline")
        except Exception as e:  # pragma: no cover
            out[f"error_{tag}"] = repr(e)


def secondary_c3_circuits(dev, out):
    """BASELINE configs[2] (Fashion-MNIST 28x28, 10 qubits, batch 1024): the two 10-qubit dense nets from the tensor
    the quantum layer receives (differN: post-PCA) through their own forward, HIP events."""
    from qiddm_amd import nn
    batch = 1024
    try:
        torch.manual_seed(42)
        net = nn.differN_noise(28, 9, 2).to(dev).eval()                     # src/fashion_ray.py:107
        red = torch.randn(batch, 10, device=dev)
        with torch.no_grad():
            us = _graph_event_us(lambda: net.forward_from_reduced(red), launches=10)
        tag = "C3_differN_noise(28,9,2)_b1024"
        out[f"denoise_images_per_s_{tag}"] = batch / (us * 1e-6)
        out[f"gate_apps_per_s_{tag}"] = batch * 900 / (us * 1e-6)
        flop = 2 * 17 * 1024 * (6 + 4 * 10)       # tangent-form layers: phase multiply + 10 RYs (4 flop) per amplitude
        out[f"roofline_{tag}"] = _valu_block(flop * batch, us, "qiddm::circuit_folded_kernel<float, 10>",
                                             "circuit + clamp(p * 784) in one launch (qiddm_forward_post), one wavefront per sample")
    except Exception as e:  # pragma: no cover
        out["error_C3_differN_noise"] = repr(e)
    try:
        torch.manual_seed(42)
        net = nn.QDenseUndirected_old_noise(60, 28).to(dev).eval()          # src/mnist_exm.py:45
        x = torch.rand(batch, 1, 28, 28, dtype=torch.double, device=dev)
        with torch.no_grad():
            us = _graph_event_us(lambda: net(x), launches=10)
        tag = "C3_QDenseUndirected_old_noise(60,28)_b1024"
        out[f"denoise_images_per_s_{tag}"] = batch / (us * 1e-6)
        out[f"gate_apps_per_s_{tag}"] = batch * 1201 / (us * 1e-6)
        # the circuit does not depend on the data: amplitude embedding -> ONE float32 product with the cached circuit unitary
        # (1024 x 2*784 operand) -> probabilities + post-processing (qiddm_amp_embed_rows / library GEMM / qiddm_prob_post)
        tf = 2.0 * 1024 * 2 * 784 * batch / (us * 1e-6) / 1e12
        out[f"roofline_{tag}"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                                  "frac": tf / VALU_PEAK_TF, "kernel": "library sgemm (hipBLASLt) between "
                                  "qiddm::amp_embed_rows_kernel and qiddm::prob_post_kernel", "kernel_avg_us": us,
                                  "note": "f32 matrix-core peak = the f32 vector peak (157.3 TFLOP/s); time of all three launches"}
        from qiddm_amd.nn import qdense as _qd
        _qd._DENSE_UNITARY = False                # the per-sample simulation kernel on the same input, for reference
        try:
            with torch.no_grad():
                us_sim = _graph_event_us(lambda: net(x), launches=10)
        finally:
            _qd._DENSE_UNITARY = True
        out[f"denoise_images_per_s_{tag}_simulated"] = batch / (us_sim * 1e-6)
        flop = 60 * 10 * 1024 * 14                # general gate path: a complex 2x2 row per amplitude and Rot (14 flop)
        out[f"roofline_{tag}_simulated"] = _valu_block(flop * batch, us_sim, "qiddm::circuit_kernel<float, 10, false>",
                                                       "amplitude embedding + 60 x (10 Rot + CNOT ring) + probs + post-processing")
    except Exception as e:  # pragma: no cover
        out["error_C3_QDenseUndirected_old_noise"] = repr(e)


def _qconv_layer_shapes(unet, batch):
    """(patch features, output channels, output pixels) of every quantum convolution in one forward of `unet` on
    `batch` 28 x 28 images.  Taken from forward hooks on a training-mode COPY run on two images (the eval-mode net
    fuses whole layer pairs into one launch and never enters QConv2d.forward) and scaled to the batch."""
    import copy
    from qiddm_amd import nn
    probe = copy.deepcopy(unet).train()
    shapes, hooks = [], []
    for m in probe.modules():
        if isinstance(m, nn.QConv2d):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, outp: shapes.append((mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1],
                                                      mod.out_channels, outp.numel() // mod.out_channels))))
    dev = next(probe.parameters()).device
    with torch.no_grad():
        probe(torch.rand(2, 1, IMG, IMG, dtype=torch.double, device=dev))
    for h in hooks:
        h.remove()
    assert shapes, "no QConv2d forward was seen"
    return [(k, c, px * batch // 2) for k, c, px in shapes]


def secondary_unet(dev, out, batch):
    """unet_simple (`UNetUndirectedS(3, 8, 3)`) as the net of the same loop: inference at the C2 batch and at C3's batch
    1024, training (Diffusion loss, backward through the seven quantum convolutions, Adam) at batch/4 x tau 10 and at
    C3's 1024 x tau 10.  The roofline blocks price the products of the circuit-unitary route
    (2 K 2C_out flop per output pixel and product: 1 forward, 4 in a training step) against the f32 MFMA peak."""
    from qiddm_amd import models, nn, noise
    from qiddm_amd.optim import FusedAdam
    from qiddm_amd.trainer import GraphedTrainStep
    for b in (batch, 1024):
        tag = f"UNetUndirectedS(3,8,3)_b{b}"
        try:
            torch.manual_seed(42)
            unet = nn.UNetUndirectedS(3, 8, 3).to(dev, dtype=torch.double).eval()
            xb = (torch.rand(b, 1, IMG, IMG, dtype=torch.double) * 0.75 + 0.5).to(dev)
            flop = sum(2.0 * k * 2 * c * px for k, c, px in _qconv_layer_shapes(unet, b))
            with torch.no_grad():
                t = _time_fn(lambda: unet(xb), 5, warm=1)
                out[f"denoise_images_per_s_{tag}"] = b / t
                g = hip_graph_recorder(lambda: unet(xb))
                tg = _time_fn(g.replay, 20, warm=2)
                out[f"denoise_images_per_s_{tag}_graphed"] = b / tg
            tf = flop / tg / 1e12
            out[f"roofline_{tag}"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                                      "frac": tf / VALU_PEAK_TF, "kernel": "qiddm::qconv_gemm_kernel (7 layers)",
                                      "step_ms": tg * 1e3,
                                      "note": "whole recorded forward (pack, GEMMs, pooling, concat, head) against the "
                                              "GEMM flop alone: a lower bound for the GEMM kernels"}
            del unet, xb, g
        except Exception as e:  # pragma: no cover
            out[f"error_{tag}"] = repr(e)
    for b in (max(batch // 4, 1), 1024):
        tag = f"UNetUndirectedS(3,8,3)_b{b}_tau10"
        try:
            torch.manual_seed(42)
            unet_t = nn.UNetUndirectedS(3, 8, 3).to(dev, dtype=torch.double).train()
            diff = models.Diffusion(unet_t, noise.add_normal_noise_multiple, "data", (IMG, IMG),
                                    torch.nn.MSELoss()).to(dev, dtype=torch.double).train()
            xt = torch.rand(b, IMG * IMG, dtype=torch.double, device=dev)
            opt = FusedAdam(diff.parameters(), lr=1e-3)

            def ustep():
                opt.zero_grad()
                diff(x=xt, T=10)
                opt.step()
            t = _time_fn(ustep, 3 if b >= 1024 else 5, warm=2)
            out[f"train_images_per_s_{tag}"] = b * 10 / t
            gstep = GraphedTrainStep(diff, opt, xt, T=10, noise="device")
            tg = _time_fn(lambda: gstep(xt), 5 if b >= 1024 else 20, warm=2)
            out[f"train_images_per_s_{tag}_graphed"] = b * 10 / tg
            flop = 4 * sum(2.0 * k * 2 * c * px for k, c, px in _qconv_layer_shapes(unet_t, b * 10))
            tf = flop / tg / 1e12
            out[f"roofline_{tag}"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                                      "frac": tf / VALU_PEAK_TF,
                                      "kernel": "qiddm::qconv_train_backward_mfma_kernel + qconv_gemm_kernel",
                                      "step_ms": tg * 1e3,
                                      "note": "whole recorded training step against the flop of the four products per "
                                              "layer (forward GEMM + three thin products of the backward)"}
            del unet_t, diff, opt, gstep, xt
            torch.cuda.empty_cache()
        except Exception as e:  # pragma: no cover
            out[f"error_{tag}"] = repr(e)


def secondary_qconv12(dev, out):
    """BASELINE config 4's layer: 12-qubit QConv2d(256 -> 256, 3x3, qdepth 3), eval mode = the one GEMM of the path
    (implicit-im2col 65536 x 2304 by 2304 x 512 on the f32 MFMA); bound: mfma.  HIP events on the launch stream
    around pack + GEMM; profiles/ holds the rocprofv3 line of qconv_gemm_wide_kernel alone."""
    from qiddm_amd import nn
    try:
        torch.manual_seed(42)
        conv = nn.QConv2d(256, 256, qdepth=3).to(dev).eval()
        xc = torch.rand(64, 256, 32, 32, dtype=torch.double, device=dev)
        with torch.no_grad():
            us = _event_time_us(lambda: conv(xc), 5, warm=2)
        px = xc.shape[0] * 32 * 32
        tf = 2.0 * 2304 * 512 * px / (us * 1e-6) / 1e12
        out["qconv12_eval_pixels_per_s"] = px / (us * 1e-6)
        out["qconv12_roofline"] = {"bound": "mfma", "achieved": tf, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": tf / VALU_PEAK_TF, "kernel": "qiddm::qconv_gemm_wide_kernel",
                                   "layer_us": us, "note": "pack + GEMM launches together (lower bound for the GEMM)"}
    except Exception as e:  # pragma: no cover
        out["qconv12_error"] = repr(e)


def secondary_flagship_training(dev, out, batch):
    """Training steps of the flagship (B * tau samples through noising + forward + backward + Adam): as written (F1: only
    linear_up trains), by the reference's declared parameter-shift rule, and by the adjoint method; eager and recorded."""
    from qiddm_amd import models, nn, noise
    x = torch.rand(batch, IMG * IMG, dtype=torch.double, device=dev)
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
            xt = x[: (batch if tag != "parameter_shift" else min(batch, 32))]
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


def secondary_measurements(dev, batch):
    """Other members of the same path -- reported next to the headline number, never instead of it (SURVEY.md
    section 8d).  Every part catches its own failure into an `error_*` key, which tools/check_bench_line.py (and the
    test suite) turn into a failed check."""
    out = {}
    secondary_dense_samplers(dev, out)
    secondary_c3_circuits(dev, out)
    secondary_unet(dev, out, batch)
    secondary_qconv12(dev, out)
    secondary_flagship_training(dev, out, batch)
    return out


def dp_train_measurement(dev, batch, world, rank):
    """The data-parallel TRAINING step of the flagship: every rank's shard of `batch` images
    x tau = 10 noise levels through the fused forward + adjoint backward, ONE flat-bucket gradient all-reduce (RCCL,
    recorded inside the step's HIP graph), one-launch Adam.  Timed like the headline (barrier, max over ranks).
    `allreduce_us`: the collective alone on the same bucket, back to back on the stream."""
    from qiddm_amd import models, nn, noise
    from qiddm_amd.optim import FusedAdam
    from qiddm_amd.trainer import GraphedTrainStep
    out = {}
    tau = 10
    torch.manual_seed(42)
    net = nn.QNN_noise(IMG * IMG, N_QUBITS, QDEPTH, detach_quantum=False)
    net.qnode.diff_method = "adjoint"
    diff = models.Diffusion(net, noise.add_normal_noise_multiple, "data", (IMG, IMG),
                            torch.nn.MSELoss()).to(dev, dtype=torch.double).train()
    torch.manual_seed(2000 + rank)                       # every rank its own shard (and its own device noise stream)
    xt = torch.rand(batch, IMG * IMG, dtype=torch.double, device=dev)
    gstep = GraphedTrainStep(diff, FusedAdam(diff.parameters(), lr=1e-3), xt, T=tau, noise="fused")
    for _ in range(5):
        gstep(xt)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    iters = 200
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        gstep(xt)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = _max_over_ranks(time.perf_counter() - t0, dev)
    out["train_images_per_s"] = world * batch * tau * iters / el
    out["train_us_per_step"] = el / iters * 1e6
    out["train_config"] = (f"QNN_noise(784,8,14), every parameter trained (adjoint), {batch} images x tau {tau} per GPU, "
                           f"dp{world}, step recorded in " + ("one HIP graph incl. the all-reduce" if gstep.g_opt is None
                                                              else "two HIP graphs around an eager all-reduce"))
    if world > 1 and gstep.bucket is not None:
        flat = next(iter(gstep.bucket.flats.values()))
        probe = torch.zeros_like(flat)
        for _ in range(10):
            dist.all_reduce(probe)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(200):
            dist.all_reduce(probe)
        torch.cuda.synchronize()
        out["allreduce_us"] = _max_over_ranks(time.perf_counter() - t0, dev) / 200 * 1e6
        out["allreduce_bytes"] = flat.numel() * flat.element_size()
    return out


def cpu_baseline(diff, x0_cpu, budget_s):
    """The oracle (CPU restatement of default.qubit's per-gate complex128 update) on the same denoise step, all
    host threads, bounded sample.  Two variants: batched (PennyLane parameter broadcasting)
    and the per-sample loop the reference's QNN_noise.forward actually runs (nn/qdense.py:278-281)."""
    from oracle import circuits as oc
    sd = {k[4:]: v.detach().cpu() for k, v in diff.state_dict().items()}

    def net(t):
        return oc.qnn_forward(t, sd["linear_down.weight"], sd["linear_down.bias"], sd["weights"],
                              sd["linear_up.weight"], sd["linear_up.bias"])

    def net_per_sample(t):
        # linear_down on the batch, then one circuit evaluation per sample in a Python loop, stacked, linear_up
        b = t.shape[0]
        xr = t.reshape(b, -1) @ sd["linear_down.weight"].T + sd["linear_down.bias"]
        spec = oc.Spec(n=N_QUBITS, encoding="rz", imprimitive="CZ", measure="expz")
        w = sd["weights"].unsqueeze(0)
        ev = torch.stack([oc.run_round(spec, xr[i], w)[0] for i in range(b)])
        return (ev @ sd["linear_up.weight"].T + sd["linear_up.bias"]).reshape(t.shape)

    res = {}
    with torch.no_grad():
        x = x0_cpu
        net(x)                                   # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            x = net(x)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 1000:
                break
        res["batched"] = (n * x.shape[0] / el, f"{n} denoise steps of the same batch-{x.shape[0]} workload in {el:.1f} s")
        xs = x0_cpu[:16]
        net_per_sample(xs[:2])
        n, t0 = 0, time.perf_counter()
        while True:
            net_per_sample(xs)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 1000:
                break
        res["per_sample_loop"] = (n * xs.shape[0] / el, f"{n} denoise steps of 16 images of the same workload, one "
                                                        f"circuit call per image, in {el:.1f} s")
    return res


def main(argv=None):
    args = parse(argv)
    world, rank, local = init_dist(args)
    dev = torch.device("cuda", local)
    from qiddm_amd import _capi
    _capi.lib()                                                  # fail loudly if the extension is missing

    diff = build_model(dev)
    torch.manual_seed(1000 + rank)
    x0 = (torch.rand(args.batch, 1, IMG, IMG, dtype=torch.double) * 0.75 + 0.5)
    spl = 1 if args.no_graph else args.steps_per_graph
    runner = Runner(diff, x0.to(dev), use_graph=not args.no_graph, spl=spl)
    elapsed, repeats = timed_region(runner, args.steps, args.warmup, world, dev)

    total_steps = args.steps * repeats
    value = world * args.batch * total_steps / elapsed
    f64 = None
    if not args.no_f64:
        try:
            f64 = f64_timed(dev, x0, args, spl, world)           # every rank: the region has barriers
        except Exception as e:  # pragma: no cover
            f64 = {"error": repr(e)}
    train = None
    if not args.no_train:
        try:
            train = dp_train_measurement(dev, args.batch, world, rank)
        except Exception as e:  # pragma: no cover - must not cost the headline line
            train = {"train_error": repr(e)}
    rank0_error = None
    if rank == 0:
        try:
            plan = launch_plan(args.steps, spl)
            kspl = max(set(plan), key=plan.count) if plan else spl     # the launch shape the timed region is made of
            kern_us = time_dominant_kernel(diff, x0.to(dev), kspl)
            circ = diff.net._circuit_descriptor()
            g_per_sample = circ.gate_count()
            alg_bytes = circ.algorithmic_bytes_per_sample("f32") * args.batch * kspl   # per launch
            hbm_eq = alg_bytes / (kern_us * 1e-6) / 1e9
            flop = lean_flop(N_QUBITS, QDEPTH, 1, IMG * IMG, False)   # what the kernel executes (tangent-form layers)
            std_flop = dense_flop(N_QUBITS, QDEPTH, 1, IMG * IMG)     # the same circuit in the (c, s) form of round 2
            valu_tf = flop * args.batch * kspl / (kern_us * 1e-6) / 1e12
            io_bytes = kspl * args.batch * IMG * IMG * 8              # one image out per step (exact); the input image is not read:
                                                                      # linear_down's output is a global phase of this circuit
            result = {
                "metric": "denoise-step images/sec, 8-qubit MNIST-28",
                "value": value,
                "unit": "images/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "repeats": repeats,
                "timed_region_s": elapsed,
                "ms_per_step": elapsed / total_steps * 1e3,
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
                           f"hipGraph replay; launches of the fused sampler hold {plan} steps for K={args.steps}; the K steps "
                           f"are repeated {repeats}x inside the timed region"
                           + (f", {Runner.GROUP} consecutive K-step launches per graph replay" if len(plan) == 1 else ""),
                           "backend": ("gloo (one-GPU rehearsal)" if args.rehearse_on_one_gpu else "nccl (RCCL)") if world > 1 else "none",
                           "parallelism": f"shard{world}"},
                "gate_apps_per_s": value * g_per_sample,
                "roofline": {
                    # The statevector of an 8-qubit sample (2 KiB) lives in the registers of four wavefronts: the kernel
                    # is bounded by instruction issue / latency on the vector ALU, not by HBM.  frac = executed VALU flop
                    # / f32 vector peak; the counters behind "latency-bound" are in profiles/ (SQ_* passes).
                    "bound": "valu", "achieved": valu_tf, "peak": VALU_PEAK_TF, "unit": "TFLOP/s",
                    "frac": valu_tf / VALU_PEAK_TF,
                    "traffic": None,
                    "traffic_profile": TRAFFIC_PROFILE,
                    "kernel": HEADLINE_KERNEL.format("float"),
                    "steps_per_launch": kspl,
                    "kernel_avg_us": kern_us,
                    "kernel_us_per_step": kern_us / kspl,
                    "executed_flop_per_sample_step": flop,
                    "standard_form_flop_per_sample_step": std_flop,
                    "achieved_standard_form": std_flop * args.batch * kspl / (kern_us * 1e-6) / 1e12,
                    "hbm_physical": {"io_bytes_per_launch": io_bytes, "GBps": io_bytes / (kern_us * 1e-6) / 1e9,
                                     "frac_of_peak": io_bytes / (kern_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "note": "images out only (exact count); weights and tables come from L2. The measured "
                                             "2*FETCH_SIZE+WRITE_SIZE per launch is in the file `traffic_profile` names "
                                             "(separate --pmc passes of this command)"},
                    "hbm_equivalent": {"bytes_per_launch": alg_bytes, "GBps": hbm_eq, "x_peak": hbm_eq / HBM_PEAK_GBS,
                                       "note": "SURVEY 8d accounting, (G+1/2)*16*2^n B per sample: what a one-sweep-per-gate "
                                               "HBM simulator would move. Not a roofline fraction: the slab never leaves "
                                               "the registers"},
                    "note": HEADLINE_NOTE,
                },
            }
            if train:
                result.update(train)
            if f64 is not None:
                result["f64"] = f64
            if world == 1:
                try:
                    result["roofline_c5"] = c5_roofline(dev)
                except Exception as e:  # pragma: no cover
                    result["roofline_c5"] = {"error": repr(e)}
            if not args.no_secondary and world == 1:
                result["secondary"] = secondary_measurements(dev, args.batch)
            if not args.no_cpu_baseline and world == 1:
                cb = cpu_baseline(diff, x0, args.cpu_seconds)
                v, sample = cb["batched"]
                vl, sample_l = cb["per_sample_loop"]
                result["cpu_baseline"] = {
                    "value": v, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                    "sample": sample + " (oracle: batched complex128 per-gate torch update)",
                    "gate_apps_per_s": v * g_per_sample,
                    "dtype": "f64",
                    "per_sample_loop": {"value": vl, "unit": "images/s", "sample": sample_l +
                                        " (the reference's own QNN_noise.forward shape, nn/qdense.py:278-281)"}}
                if f64 is not None and "images_per_s" in f64:
                    # like for like: the reference's precision on both sides, both from a timed region
                    result["speedup_vs_cpu_same_precision"] = f64["images_per_s"] / v
                result["speedup_vs_cpu_f32_gpu"] = value / v
            print(json.dumps(result), flush=True)

        except BaseException as e:   # the other ranks wait in the barrier below: reach it, then re-raise
            rank0_error = e
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank0_error is not None:
        raise rank0_error


if __name__ == "__main__":
    main()
