#!/usr/bin/env python3
"""Group the design-Q dispatches of a rocprofv3 --kernel-trace of `bench.py` into bursts (a burst ends where the GPU idles for 30 us) and, per burst: how
many dispatches, on which hardware queues, span / dispatches, interval between starts, the kernel's own duration, and the HBM-roofline fraction the span
gives (127 795 200 algorithmic bytes per call of BASELINE configs[2]).  usage: trace_bursts.py kernel_trace.csv out.json [bench.json]"""
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_mfir" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"])) for r in rows]
bursts, cur = [], [k[0]]
for a in k[1:]:
    if a[0] - max(e for _, e, _ in cur) > 30000:
        bursts.append(cur); cur = [a]
    else:
        cur.append(a)
bursts.append(cur)
ALG = 127795200.0
out = {"kernel": rows[0]["Kernel_Name"], "bursts": []}
for b in bursts:
    qs = sorted(set(q for _, _, q in b)); span = max(e for _, e, _ in b) - b[0][0]
    dur = [e - s for s, e, _ in b]; st = [s for s, _, _ in b]
    iv = [st[i + 1] - st[i] for i in range(len(st) - 1)] or [0]
    out["bursts"].append({"dispatches": len(b), "queues": len(qs), "us_span_per_call": round(span / len(b) / 1e3, 2), "us_start_to_start_median": round(sorted(iv)[len(iv) // 2] / 1e3, 2),
                          "us_start_to_start_first8": [round(x / 1e3, 1) for x in iv[:8]], "us_kernel_duration_mean": round(sum(dur) / len(dur) / 1e3, 2),
                          "us_kernel_duration_first_last": [round(dur[0] / 1e3, 1), round(dur[-1] / 1e3, 1)], "frac_of_8TBs_from_span": round(ALG / (span / len(b) * 1e-9) / 8e12, 4)})
if len(sys.argv) > 3:
    try:
        r = json.load(open(sys.argv[3])); out["bench_line"] = {"value": r["value"], "ms_per_step": r["ms_per_step"], "frac": r["roofline"]["frac"], "frac_sustained": r["roofline"].get("frac_sustained"),
                                                              "overlapped": r["roofline"].get("overlapped_calls")}
    except Exception as e:
        out["bench_line"] = str(e)
json.dump(out, open(sys.argv[2], "w"), indent=1)
for b in out["bursts"]:
    print(json.dumps(b))
print(json.dumps(out.get("bench_line")))
