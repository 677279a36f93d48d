#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
CLS=${1:-mixed:10}; OUT=$PWD/gpurun_out/r05_mixtrace; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp && timeout 120 rocprofv3 --output-format csv --kernel-trace -d "$OUT" -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-steady --iq-class $CLS > "$OUT/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 - "$(find $OUT -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("k_mfir" in r["Kernel_Name"] or "k_fastb" in r["Kernel_Name"] or "k_stream" in r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the first long burst with 2+ queues and fastb kernels after the first 120 calls
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), "B" if "fastb" in r["Kernel_Name"] else ("S" if "k_stream" in r["Kernel_Name"] else "Q"), int(r["Grid_Size"]) if "Grid_Size" in r else 0) for r in rows]
nb = sum(1 for x in k if x[3] == "B")
print("dispatches", len(k), "of them design B", nb)
# print 40 consecutive dispatches from the middle of the trace where B kernels appear
idx = [i for i, x in enumerate(k) if x[3] == "B"]
import collections
for frac in (0.15, 0.5):
    i0 = idx[int(len(idx) * frac)] if idx else 0
    t0 = k[i0][0]
    print("---- from dispatch", i0)
    for x in k[i0:i0 + 36]:
        print("%s start %8.1f end %8.1f dur %6.1f queue %d grid %d" % (x[3], (x[0] - t0) / 1e3, (x[1] - t0) / 1e3, (x[1] - x[0]) / 1e3, x[2], x[4]))
PY
