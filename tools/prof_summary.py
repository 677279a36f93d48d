#!/usr/bin/env python3
"""Condense rocprofv3 output directories (kernel-trace stats + PMC passes) into one JSON summary per run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(out):
    summary = {"dir": os.path.basename(out), "kernels": {}, "pmc": {}}
    for fn in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            summary["kernels"][row["Name"][:90]] = {k: row[k] for k in row if k != "Name"}
    for fn in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(lambda: defaultdict(int))
        for row in csv.DictReader(open(fn)):
            k = row.get("Kernel_Name", "?")[:90]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
        for k in acc:
            d = summary["pmc"].setdefault(k, {})
            for c in acc[k]:
                d[c] = {"mean_per_dispatch": acc[k][c] / cnt[k][c], "dispatches": cnt[k][c]}
    print(json.dumps(summary, indent=1))
    with open(os.path.join(out, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
