#!/usr/bin/env python3
"""Turn the rocprofv3 PMC csv files of tools/profile_round.sh into <tag>_pmc_summary.json and traffic_<tag>.json.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in separate
passes (TCC slots), are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
read (16 B per lane: this kernel's buffer_load_dwordx4 stream), so the read side is doubled.  WRITE_SIZE is uncalibrated
on gfx950 and taken as is (it is 4 % of the traffic here)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(work, out, tag):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for fn in glob.glob(os.path.join(work, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
    summary = {}
    for k in acc:
        summary[k] = {c: {"mean_per_dispatch": acc[k][c] / cnt[k][c], "dispatches": cnt[k][c]} for c in sorted(acc[k])}
    with open(os.path.join(out, "%s_pmc_summary.json" % tag), "w") as f:
        json.dump(summary, f, indent=1)
    kern = None
    for k in summary:
        if ("k_fast" in k or "k_generic" in k or "k_stream" in k) and "FETCH_SIZE" in summary[k]:
            if kern is None or summary[k]["FETCH_SIZE"]["mean_per_dispatch"] > summary[kern]["FETCH_SIZE"]["mean_per_dispatch"]:
                kern = k
    bench = {}
    try:
        bench = json.loads(open(os.path.join(out, "%s_bench_under_rocprof.json" % tag)).read())
    except Exception:
        pass
    if kern:
        fetch_kib = summary[kern]["FETCH_SIZE"]["mean_per_dispatch"]
        write_kib = summary[kern].get("WRITE_SIZE", {}).get("mean_per_dispatch", 0.0)
        t = {
            "round": tag, "rocprof_kernel": kern,
            "kernel_name": bench.get("config", {}).get("kernel"),
            "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
            "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane streams); write bytes = WRITE_SIZE x 1024",
            "hbm_read_bytes_per_launch": 2.0 * fetch_kib * 1024.0,
            "hbm_write_bytes_per_launch": write_kib * 1024.0,
            "hbm_bytes_per_launch": 2.0 * fetch_kib * 1024.0 + write_kib * 1024.0,
            "algorithmic_bytes_per_launch": bench.get("roofline", {}).get("algorithmic_bytes_per_launch"),
        }
        with open(os.path.join(out, "traffic_%s.json" % tag), "w") as f:
            json.dump(t, f, indent=1)
        print(json.dumps(t))
    print(open(os.path.join(out, "%s_kernel_stats.csv" % tag)).read()[:1500])


if __name__ == "__main__":
    main(*sys.argv[1:4])
