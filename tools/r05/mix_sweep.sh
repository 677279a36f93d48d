#!/bin/bash
# tools/r05/mix_sweep.sh — the one-launch kernel's balance knobs (development library): tile R of the design-B workgroups, workgroups per CU, the cost
# factor that sets the shares, the minimum sub-tiles of a design-B segment
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
run "fm" fm A=1
for cls in mixed:10 mixed:25; do
  run "R4 default" $cls A=1
  run "R4 cost 2.7" $cls SDRFM_MIX_COST=2.7
  run "R4 cost 3.5" $cls SDRFM_MIX_COST=3.5
  run "R4 cost 2.7 w14" $cls SDRFM_MIX_COST=2.7 SDRFM_MIX_WAVES_PER_CU=14
  run "R4 cost 2.7 w13" $cls SDRFM_MIX_COST=2.7 SDRFM_MIX_WAVES_PER_CU=13
  run "R4 cost 2.7 w12" $cls SDRFM_MIX_COST=2.7 SDRFM_MIX_WAVES_PER_CU=12
  run "R4 cost 2.7 ms2" $cls SDRFM_MIX_COST=2.7 SDRFM_MIN_SUBTILES=2
  run "R8 cost 2.0" $cls SDRFM_MIX_R=8
  run "R8 cost 2.5" $cls SDRFM_MIX_R=8 SDRFM_MIX_COST=2.5
  run "R8 cost 2.5 ms2" $cls SDRFM_MIX_R=8 SDRFM_MIX_COST=2.5 SDRFM_MIN_SUBTILES=2
  run "R8 cost 3.0 ms2" $cls SDRFM_MIX_R=8 SDRFM_MIX_COST=3.0 SDRFM_MIN_SUBTILES=2
done
run "fm" fm A=1
} | tee $OUT/sweep_$(date +%H%M%S).txt
