#!/bin/bash
# tools/r05/final_check.sh — what the driver runs at the end of a round, on one box: the GPU tests, smoke(), the default bench line and the driver-style one
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 300 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$OUT/bench_err.txt | tail -1 > $OUT/bench_steps_20_warmup_5.json
python3 - $OUT/bench_steps_20_warmup_5.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]; o = rf.get("overlapped_calls", {})
print("value %.4g %s | serial frac %.4f sustained %.4f steady %s | overlapped frac %.4f sustained %.4f steady %s | cpu %s" % (r["value"], r["unit"], rf["frac"], rf.get("frac_sustained", 0),
      rf.get("frac_steady"), o.get("frac", 0), o.get("frac_sustained", 0), o.get("frac_steady"), r.get("cpu_baseline", {}).get("value")))
PY
