#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace of `bench.py` (fm workload, default = timed calls made with SDRFM_F_OVERLAP): the design-Q
dispatches split into the overlapped burst (two hardware queues taken in turn) and the serial bursts (one queue), with the per-call
interval (start to start), the kernel's own duration and how much of the burst had two kernels resident.
usage: overlap_trace_summarize.py kernel_trace.csv out.json [commit]"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_mfir" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])) for r in rows]
# bursts: dispatches back to back or overlapping
bursts, cur = [], [k[0]]
for a in k[1:]:
    if a[0] - max(e for _, e, _ in cur[-2:]) > 30000:        # idle for 30 us: the bench is between two timed regions
        bursts.append(cur); cur = [a]
    else:
        cur.append(a)
bursts.append(cur)
out = {"source": "rocprofv3 --kernel-trace -- python3 bench.py --steps 100 --warmup 10 (design-Q dispatches only)", "commit": sys.argv[3] if len(sys.argv) > 3 else None,
       "kernel": rows[0]["Kernel_Name"], "bursts": []}
for b in bursts:
    if len(b) < 20:
        continue
    queues = sorted(set(q for _, _, q in b))
    span = b[-1][1] - b[0][0]
    dur = [e - s for s, e, _ in b]
    starts = [s for s, _, _ in b]
    gaps = [starts[i + 1] - starts[i] for i in range(len(starts) - 1)]
    # time with >= 2 kernels resident
    ev = sorted([(s, 1) for s, _, _ in b] + [(e, -1) for _, e, _ in b])
    two, n, last = 0, 0, ev[0][0]
    for t, d in ev:
        if n >= 2:
            two += t - last
        n += d; last = t
    out["bursts"].append({"dispatches": len(b), "hardware_queues": queues, "mode": "overlapped (two queues in turn)" if len(queues) > 1 else "serial (one queue)",
                          "us_per_call_span": round(span / len(b) / 1e3, 3), "us_start_to_start_mean": round(sum(gaps) / len(gaps) / 1e3, 3),
                          "us_kernel_duration_mean": round(sum(dur) / len(dur) / 1e3, 3), "fraction_of_span_with_two_kernels_resident": round(two / span, 3)})
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
