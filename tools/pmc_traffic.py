#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes written by tools/profile_bench.sh to
HBM bytes per launch for every kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md
("HBM") prescribes for gfx950:

    FETCH_SIZE, WRITE_SIZE are reported in KiB;
    FETCH_SIZE counts a wide coalesced streaming read at exactly 1/2 of its bytes -> x2;
    WRITE_SIZE is exact for streaming stores.

    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        (per launch: mean over dispatches)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    out = sys.argv[1]
    batch = int(os.environ.get("QIDDM_BENCH_BATCH", "256"))
    rd = collect(os.path.join(out, "pmc_rd"), "FETCH_SIZE")
    wr = collect(os.path.join(out, "pmc_wr"), "WRITE_SIZE")
    recs = []
    for k in sorted(set(rd) | set(wr)):
        f = sum(rd.get(k, [0.0])) / max(len(rd.get(k, [])), 1)
        w = sum(wr.get(k, [0.0])) / max(len(wr.get(k, [])), 1)
        recs.append({"kernel": k, "batch": batch, "dispatches": len(rd.get(k, [])),
                     "fetch_size_kib_mean": f, "write_size_kib_mean": w,
                     "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0})
    json.dump({"correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE halves wide reads)",
               "kernels": recs}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
