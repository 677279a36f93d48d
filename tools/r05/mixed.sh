#!/bin/bash
# tools/r05/mixed.sh — VERDICT r04 item 2: bench.py on batches that mix carriers and noise-only streams (per-stream routing), beside the all-carrier batch, one box
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_mixed; mkdir -p $OUT
for rep in 1 2 3; do
  for cls in fm mixed:5 mixed:10 mixed:25 random; do
    timeout 90 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-steady --iq-class $cls 2>/dev/null | tail -1 > $OUT/last.json || echo "TIMEOUT $cls rep $rep"
    python3 - $OUT/last.json $cls $rep <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]; o = rf.get("overlapped_calls", {}); rt = r.get("routing", {})
print("%-9s rep %s  serial %.2f us (sustained %.2f) | overlapped %.2f us (sustained %.2f) | on bit-exact kernels %s | %s" % (
    sys.argv[2], sys.argv[3], rf["kernel_ms_avg"] * 1e3, rf.get("kernel_ms_sustained", 0) * 1e3, o.get("ms_per_call", 0) * 1e3,
    127795200 / (o.get("frac_sustained", 1) * 8e12) * 1e6 if o else 0, rt.get("streams_on_bit_exact_kernels"), rt.get("kernels_last_call", r["config"]["kernel"])))
PY
  done
done | tee $OUT/mixed_$(date +%H%M%S).txt
