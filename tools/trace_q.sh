#!/bin/bash
# tools/trace_q.sh <tag> [bench args]: rocprofv3 kernel trace of bench.py; prints per-kernel stats and the launch-to-launch cadence of k_mfir
TAG=${1:-q}; shift || true
OUT=$PWD/gpurun_out/trace_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT" -o trace -- python3 bench.py --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
grep '^{"metric"' "$OUT/bench.log" | tail -1 > "$OUT/bench.json"
find "$OUT" -name "*kernel_stats.csv" -exec cat {} \; | head -8
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
import numpy as np
rows = []
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_mfir" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
s = np.array([a for a, b in rows]); e = np.array([b for a, b in rows])
dur = (e - s) / 1e3
gap = (s[1:] - e[:-1]) / 1e3
cad = (s[1:] - s[:-1]) / 1e3
sel = slice(len(dur) // 2, None)      # second half: sustained clocks
print("k_mfir launches %d: duration us mean %.2f (second half %.2f), start-to-start %.2f (second half %.2f), gap end->next start %.2f (second half %.2f)" % (
    len(dur), dur.mean(), dur[sel].mean(), np.median(cad), np.median(cad[len(cad) // 2:]), np.median(gap), np.median(gap[len(gap) // 2:])))
PY
