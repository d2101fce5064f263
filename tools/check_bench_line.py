#!/usr/bin/env python
"""Gate on a bench.py JSON line:  python bench.py ... | python tools/check_bench_line.py [--require KEY ...]

Exits non-zero when the input holds no JSON line, when a contract field is missing, or when ANY key anywhere in
the line contains "error" (every part of bench.py catches its own failure into such a key so that one broken secondary
cannot cost the headline -- this checker is what makes that failure loud).  Prints the offending paths."""
from __future__ import annotations

import argparse
import json
import sys

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")
ROOFLINE = ("bound", "achieved", "peak", "unit", "frac", "traffic")


def error_paths(node, path=""):
    """Paths of every key containing 'error' (case-insensitive) and of every non-finite number."""
    bad = []
    if isinstance(node, dict):
        for k, v in node.items():
            here = f"{path}.{k}" if path else str(k)
            if "error" in str(k).lower():
                bad.append(f"{here} = {v!r}")
            bad += error_paths(v, here)
    elif isinstance(node, (list, tuple)):
        for i, v in enumerate(node):
            bad += error_paths(v, f"{path}[{i}]")
    elif isinstance(node, float) and (node != node or node in (float("inf"), float("-inf"))):
        bad.append(f"{path} = {node!r} (not finite)")
    return bad


def last_json_line(text: str):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                return json.loads(line)
            except json.JSONDecodeError:
                continue
    return None


def check(line: dict, require=(), single_gpu_extras=True):
    problems = error_paths(line)
    problems += [f"missing contract field {k!r}" for k in CONTRACT if k not in line]
    problems += [f"missing roofline.{k}" for k in ROOFLINE if k not in line.get("roofline", {})]
    if single_gpu_extras and line.get("n_gpus") == 1:
        problems += [f"missing {k!r} (N = 1 line)" for k in ("cpu_baseline",) if k not in line]
    for key in require:
        node = line
        for part in key.split("."):
            if not isinstance(node, dict) or part not in node:
                problems.append(f"required key {key!r} is missing")
                break
            node = node[part]
    if isinstance(line.get("value"), (int, float)) and not line["value"] > 0:
        problems.append(f"value = {line['value']!r} is not positive")
    return problems


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("file", nargs="?", help="file holding the line (default: stdin)")
    ap.add_argument("--require", action="append", default=[], help="dotted key that must be present (repeatable)")
    ap.add_argument("--echo", action="store_true", help="print the line itself to stdout as well")
    a = ap.parse_args(argv)
    text = open(a.file).read() if a.file else sys.stdin.read()
    line = last_json_line(text)
    if line is None:
        print("[check_bench_line] no JSON line in the input", file=sys.stderr)
        return 2
    if a.echo:
        print(json.dumps(line))
    problems = check(line, a.require)
    for p in problems:
        print(f"[check_bench_line] {p}", file=sys.stderr)
    if not problems:
        print(f"[check_bench_line] ok: {line['metric']} = {line['value']:.6g} {line['unit']} on {line['n_gpus']} GPU(s)",
              file=sys.stderr)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
