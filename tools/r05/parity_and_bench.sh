#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_soak_head; mkdir -p $OUT
for i in 1 2 3; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-steady 2>/dev/null | tail -1 > $OUT/b.json
  python3 - $OUT/b.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]; o = rf["overlapped_calls"]
print("value %.4g ms_per_step %.4f | serial frac %.4f | overlapped frac %.4f (%.2f us) of_measured %s" % (r["value"], r["ms_per_step"], rf["frac"], o["frac"], o["ms_per_call"] * 1e3, o.get("frac_of_measured")))
PY
done
timeout 700 python tools/fuzz_parity.py 240 9201 > $OUT/fuzz_parity_full.txt 2>&1; tail -12 $OUT/fuzz_parity_full.txt | cut -c1-300
