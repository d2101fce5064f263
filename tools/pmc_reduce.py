#!/usr/bin/env python3
"""Mean per launch of every counter in a rocprofv3 --pmc output directory, grouped by kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = []
    for k in sorted(acc):
        rec = {"kernel": k, "dispatches": max(len(v) for v in acc[k].values())}
        for c, v in sorted(acc[k].items()):
            rec[c] = sum(v) / len(v)
        out.append(rec)
    json.dump({"per_launch_mean": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
