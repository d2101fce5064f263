"""CPU tests of bench.py's own machinery: the launch plan / Runner bookkeeping (the round-2 driver line lost two
secondaries to `prepare(75); run(150)`), the N-rank self-launch command, and the line checker that turns any
`error_*` key into a non-zero exit."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bench  # noqa: E402
import check_bench_line  # noqa: E402


class FakeDiff:
    """denoise_steps(x, m): every step adds 1 to the image; counts launches and steps."""

    def __init__(self):
        self.launches, self.steps = 0, 0

    def denoise_steps(self, x, m):
        self.launches += 1
        self.steps += m
        return torch.stack([x + (i + 1) for i in range(m)])


class FakeGraph:
    def __init__(self, fn):
        self.fn, self.replays = fn, 0

    def replay(self):
        self.replays += 1
        self.fn()


def fake_recorder(fn):
    fn()                     # the real recorder runs the chain once on a side stream and once under capture ...
    fn()
    return FakeGraph(fn)


def make_runner(spl=15, lpg=8):
    diff = FakeDiff()
    r = bench.Runner(diff, torch.zeros(2, 1, 4, 4), True, spl, launches_per_graph=lpg, recorder=fake_recorder)
    return diff, r


@pytest.mark.parametrize("k", [0, 1, 5, 14, 15, 16, 20, 29, 30, 75, 150, 2000])
def test_launch_plan_covers_exactly_k_steps(k):
    plan = bench.launch_plan(k, 15)
    assert sum(plan) == k
    assert all(15 <= m < 30 for m in plan) or k < 15
    assert plan == ([] if k == 0 else plan)


@pytest.mark.parametrize("k", [5, 20, 75, 150, 2000])
def test_runner_chunks_partition_the_plan(k):
    _, r = make_runner()
    chunks = r._chunks(k)
    assert [m for c in chunks for m in c] == bench.launch_plan(k, 15)
    assert all(1 <= len(c) <= r.lpg for c in chunks)


def test_prepare_75_then_run_150_is_exactly_150_steps():
    """The round-2 regression: a step count whose chunks were never recorded must still run (recorded on first use)."""
    diff, r = make_runner()
    r.prepare(75)
    recorded = set(r.graphs)
    assert recorded == {(15,) * 5}
    x_before, steps_before = r.x.clone(), diff.steps
    r.run(150)                                   # chunks (15,)*8 and (15,)*2: neither prepared
    assert set(r.graphs) == recorded | {(15,) * 8, (15,) * 2}
    assert r.lazy_records == 2
    # every recording runs its chain twice (warm + capture stand-ins), then the replay once
    assert r.steps_done == 150
    lazily_run = 2 * (15 * 8 + 15 * 2)
    assert diff.steps - steps_before == 150 + lazily_run
    assert torch.equal(r.x, x_before + 150 + lazily_run)
    # a second run replays only
    steps_before = diff.steps
    r.run(150)
    assert diff.steps - steps_before == 150 and r.lazy_records == 2


@pytest.mark.parametrize("k", [5, 20, 2000])
def test_prepared_runs_record_nothing_in_the_timed_region(k):
    diff, r = make_runner()
    r.prepare(k)
    r.prepare_repeated(k)
    n_graphs = len(r.graphs)
    for reps in (1, 7, 8, 9, 17):
        before, done = diff.steps, r.steps_done
        r.run_repeated(k, reps)
        assert diff.steps - before == reps * k, (k, reps)
        assert r.steps_done - done == reps * k
    assert len(r.graphs) == n_graphs and r.lazy_records == 0


def test_run_repeated_groups_eight_single_launch_blocks_per_replay():
    diff, r = make_runner()
    r.prepare(20)
    r.prepare_repeated(20)                       # K = 20 is ONE launch of 20 steps -> a graph of 8 such launches
    key = (20,) * bench.Runner.GROUP
    assert key in r.graphs
    group, single = r.graphs[key], r.graphs[(20,)]
    g0, s0 = group.replays, single.replays
    r.run_repeated(20, 19)
    assert group.replays - g0 == 2 and single.replays - s0 == 3
    # K = 2000 is many launches: no group graph, plain chunked replays
    r.prepare_repeated(2000)
    assert all(len(set(k)) <= 2 for k in r.graphs)
    assert tuple(bench.launch_plan(2000, 15)) * bench.Runner.GROUP not in r.graphs


def test_eager_runner_steps_one_at_a_time():
    diff = FakeDiff()
    r = bench.Runner(diff, torch.zeros(1, 1, 2, 2), False, 1)
    r.prepare(7)
    r.run(7)
    assert diff.launches == 7 and diff.steps == 7 and not r.graphs


def test_dense_flop_matches_the_headline_count():
    # 13 simulated layers x 256 amplitudes x (6 + 6*8) + generated first layer + <Z> + the two linears
    assert bench.dense_flop(8, 14, 1, 784) == 13 * 256 * 54 + 8 * 256 + 2 * 8 * 256 + 2 * 2 * 784 * 8 == 210944


def test_lean_flop_is_the_tangent_form_count():
    # 13 simulated layers x 256 amplitudes x (6 + 8 * 4) + read-out 256 * (3 + 8) + linear_up 2 * 784 * 8
    assert bench.lean_flop(8, 14, 1, 784, False) == 13 * 256 * 38 + 256 * 11 + 2 * 784 * 8 == 141824
    assert bench.lean_flop(8, 12, 2, 784, True) == 2 * (11 * 256 * 38 + 256 * 11) + 2 * 784 * 8 + 128
    assert bench.lean_flop(8, 14, 1, 784, False) < bench.dense_flop(8, 14, 1, 784)


# ---- the N-rank self-launch -----------------------------------------------------------------------------------------
def _dry(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--dry-run-launch"],
                       capture_output=True, text=True, env=e, timeout=120)
    assert p.returncode == 0, p.stderr
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_gpus_n_without_a_launcher_starts_n_ranks_on_loopback():
    d = _dry("--gpus", "4", "--steps", "20", "--warmup", "5")
    cmd = d["command"]
    assert d["self_launch"] is True
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]     # arguments forwarded verbatim
    assert "--dry-run-launch" not in cmd


def test_no_self_launch_under_a_launcher_or_on_one_gpu():
    assert _dry("--gpus", "1")["self_launch"] is False
    assert _dry()["self_launch"] is False
    assert _dry("--gpus", "8", env={"WORLD_SIZE": "8", "RANK": "0"})["self_launch"] is False


def test_world_size_must_match_gpus(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.init_dist(bench.parse(["--gpus", "8"]))
    assert "WORLD_SIZE=2" in str(e.value)


# ---- the line checker -------------------------------------------------------------------------------------------
GOOD = {"metric": "m", "value": 1.0, "unit": "images/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.1,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "w"}, "roofline": {"bound": "valu", "achieved": 1.0, "peak": 2.0, "unit": "TFLOP/s",
                                                  "frac": 0.5, "traffic": None},
        "cpu_baseline": {"value": 1.0, "unit": "images/s", "cores": 8, "kind": "port", "sample": "s"},
        "secondary": {"denoise_images_per_s_X": 3.0}}


def test_checker_accepts_a_clean_line_and_finds_nested_errors(tmp_path):
    assert check_bench_line.check(GOOD) == []
    bad = json.loads(json.dumps(GOOD))
    bad["secondary"]["error_QIDDM_LL_noise(784,8,6,2)"] = "KeyError((15,)*8)"     # the round-2 line
    bad["f64"] = {"error": "boom"}
    bad["secondary"]["unet_train_error"] = "x"
    problems = check_bench_line.check(bad)
    assert len(problems) == 3 and all("error" in p for p in problems)
    f = tmp_path / "line.json"
    f.write_text("noise\n" + json.dumps(bad) + "\n")
    assert check_bench_line.main([str(f)]) == 1
    f.write_text(json.dumps(GOOD) + "\n")
    assert check_bench_line.main([str(f)]) == 0
    assert check_bench_line.main([str(f), "--require", "secondary.denoise_images_per_s_X"]) == 0
    assert check_bench_line.main([str(f), "--require", "secondary.nope"]) == 1
    f.write_text("no json here\n")
    assert check_bench_line.main([str(f)]) == 2


def test_checker_flags_missing_contract_fields_and_nan():
    line = json.loads(json.dumps(GOOD))
    del line["roofline"]["traffic"]
    del line["cpu_baseline"]
    line["value"] = float("nan")
    problems = check_bench_line.check(line)
    assert any("roofline.traffic" in p for p in problems)
    assert any("cpu_baseline" in p for p in problems)
    assert any("not finite" in p for p in problems)
