#!/bin/bash
# tools/r05/ab_bench.sh <libA.so> <libB.so> [reps] [bench args...] — bench.py with two builds of the library in turn on ONE box (the product library is swapped in
# place: csrc/libsdrfm.so); prints the roofline fractions of every run.  libA / libB are file names under stm32f7-rtlsdr_amd/csrc.
cd "$GRAFT_REPO_ROOT" || exit 1
CS=stm32f7-rtlsdr_amd/csrc; A=$1; B=$2; REPS=${3:-5}; shift 3
OUT=gpurun_out/r05_ab_bench; mkdir -p $OUT
cp $CS/libsdrfm.so $CS/libsdrfm_keep.so
for rep in $(seq 1 $REPS); do
  for v in $A $B; do
    cp $CS/$v $CS/libsdrfm.so
    python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/last.json
    python3 - $OUT/last.json $v $rep <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]; o = rf.get("overlapped_calls", {})
print("%-22s rep %s  serial frac %.4f (%.2f us) sustained %.4f | overlapped frac %.4f (%.2f us) sustained %.4f | peak_measured %.0f" % (
    sys.argv[2], sys.argv[3], rf["frac"], rf["kernel_ms_avg"] * 1e3, rf.get("frac_sustained", 0), o.get("frac", 0), o.get("ms_per_call", 0) * 1e3, o.get("frac_sustained", 0), rf.get("peak_measured", 0)))
PY
  done
done | tee $OUT/ab_$(date +%H%M%S).txt
cp $CS/libsdrfm_keep.so $CS/libsdrfm.so
