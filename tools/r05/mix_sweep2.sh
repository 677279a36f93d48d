#!/bin/bash
# tools/r05/mix_sweep2.sh — second pass over the one-launch kernel's two knobs that mattered, three alternating repetitions
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_mix_sweep; mkdir -p $OUT
run() {  # name cls env...
  local name=$1 cls=$2; shift 2
  env "$@" timeout 90 python bench.py --dev-library --steps 100 --warmup 10 --no-cpu-baseline --no-steady --iq-class $cls 2>/dev/null | tail -1 > $OUT/last.json || echo "TIMEOUT"
  python3 - $OUT/last.json "$name" $cls <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]; o = rf.get("overlapped_calls", {}); rt = r.get("routing", {})
print("%-34s %-9s serial %.2f us (sus %.2f) | overlapped %.2f us (sus %.2f) | %s" % (sys.argv[2], sys.argv[3], rf["kernel_ms_avg"] * 1e3, rf.get("kernel_ms_sustained", 0) * 1e3,
      o.get("ms_per_call", 0) * 1e3, 127795200 / (o.get("frac_sustained", 1) * 8e12) * 1e6 if o else 0, rt.get("streams_on_bit_exact_kernels")))
PY
}
{
for rep in 1 2 3; do
run "fm" fm A=1
for cls in mixed:10 mixed:25; do
  run "cost 2.0 w15" $cls A=1
  run "cost 2.7 w15" $cls SDRFM_MIX_COST=2.7
  run "cost 2.0 w12" $cls SDRFM_MIX_WAVES_PER_CU=12
  run "cost 2.7 w12" $cls SDRFM_MIX_COST=2.7 SDRFM_MIX_WAVES_PER_CU=12
  run "cost 3.2 w12" $cls SDRFM_MIX_COST=3.2 SDRFM_MIX_WAVES_PER_CU=12
  run "two launches" $cls SDRFM_MIX_OFF=1
done
done
} | tee $OUT/sweep2_$(date +%H%M%S).txt
