#!/bin/bash
# tools/r05/driver_trace.sh <tag> — rocprofv3 kernel trace of the driver's exact command; the design-Q dispatches grouped by burst
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-x}; OUT=$PWD/gpurun_out/r05_drvtrace_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT" -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"
grep '^{"metric"' "$OUT/bench.log" | tail -1 > "$OUT/bench.json"
python3 tools/r05/trace_bursts.py "$(find $OUT -name '*kernel_trace.csv' | head -1)" "$OUT/bursts.json" "$OUT/bench.json"
find "$OUT" -name "*kernel_stats.csv" -exec head -6 {} \;
