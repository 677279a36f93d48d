#!/bin/bash
# tools/r05/mix1.sh — the one-launch kernel of a mixed batch (k_mix): routing tests, then bench.py on mixed batches with the product library
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_mix1; mkdir -p $OUT
timeout 600 python -m pytest tests/test_route_gpu.py -x -q --timeout 200 2>&1 | tail -15 > $OUT/pytest.txt
cat $OUT/pytest.txt
for rep in 1 2; do
  for cls in fm mixed:10 mixed:25; do
    timeout 90 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-steady --iq-class $cls 2>$OUT/err_$cls.txt | tail -1 > $OUT/last.json || echo "TIMEOUT $cls rep $rep"
    python3 - $OUT/last.json $cls $rep <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]; o = rf.get("overlapped_calls", {}); rt = r.get("routing", {})
print("%-9s rep %s  serial %.2f us (sustained %.2f) | overlapped %.2f us (sustained %.2f) | on bit-exact kernels %s | %s" % (
    sys.argv[2], sys.argv[3], rf["kernel_ms_avg"] * 1e3, rf.get("kernel_ms_sustained", 0) * 1e3, o.get("ms_per_call", 0) * 1e3,
    127795200 / (o.get("frac_sustained", 1) * 8e12) * 1e6 if o else 0, rt.get("streams_on_bit_exact_kernels"), rt.get("kernels_timed_region", r["config"]["kernel"])))
PY
  done
done | tee $OUT/mixed_$(date +%H%M%S).txt
