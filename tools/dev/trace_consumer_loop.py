#!/usr/bin/env python3
"""tools/dev/trace_consumer_loop.py <kernel_trace.csv> — what the kernel trace of the consumer loop's fast form says: for the demodulator's kernels (k_mfir) the interval
between consecutive starts and their durations, per hardware queue; for the sink's kernels (k_pcm_sink_scan) their duration, which demodulator kernels run while one does,
and whether a demodulator kernel's start follows a sink kernel's end (the two taking turns on one queue / pipe).  Development aid (profiles/r06_sink.txt)."""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
k = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
k.sort(key=lambda x: x[1])
dem = [x for x in k if "k_mfir" in x[0]][-600:]
snk = [x for x in k if "k_pcm_sink_scan" in x[0]][-600:]
if not dem or not snk:
    print("kernels found:", {x[0][:40] for x in k}); sys.exit(0)
d_int = [(b[1] - a[1]) / 1e3 for a, b in zip(dem, dem[1:])]
print("demodulator: %d kernels on queues %s; start-to-start median %.2f us (mean %.2f); duration median %.2f us" % (len(dem), sorted({x[3] for x in dem}), st.median(d_int), st.mean(d_int), st.median([(x[2] - x[1]) / 1e3 for x in dem])))
print("sink: %d kernels on queues %s; duration median %.2f us (min %.2f max %.2f)" % (len(snk), sorted({x[3] for x in snk}), st.median([(x[2] - x[1]) / 1e3 for x in snk]), min((x[2] - x[1]) / 1e3 for x in snk), max((x[2] - x[1]) / 1e3 for x in snk)))
# while a sink kernel runs: how many demodulator kernels are resident, and how much demodulator time the sink's span overlaps
res = []
for s_ in snk:
    n = sum(1 for d in dem if d[1] < s_[2] and d[2] > s_[1])
    res.append(n)
print("demodulator kernels resident while a sink kernel runs: min %d median %d max %d" % (min(res), st.median(res), max(res)))
# does a demodulator kernel start right behind a sink kernel's end (taking turns)?
gaps = []
for s_ in snk:
    nxt = [d[1] for d in dem if d[1] >= s_[2]]
    if nxt: gaps.append((min(nxt) - s_[2]) / 1e3)
print("first demodulator start behind a sink kernel's end: median %.2f us (min %.2f)" % (st.median(gaps), min(gaps)))
# per-queue: is each demod queue idle while the sink runs?
for q in sorted({x[3] for x in dem}):
    dq = [x for x in dem if x[3] == q]
    idle = [(b[1] - a[2]) / 1e3 for a, b in zip(dq, dq[1:])]
    print("queue %s: %d kernels, gap between one's end and the next's start: median %.2f us (p90 %.2f)" % (q, len(dq), st.median(idle), sorted(idle)[int(0.9 * len(idle))]))
