#!/bin/bash
# tools/r05/nvt.sh — design B alone (bit-exact handle, development library): channel taps in SGPRs (12 pairs in VGPRs, 20 in SGPRs: mode 0) against all 32 pairs in VGPRs (mode 1 / 2: a build with every tap pair in VGPRs — measured, not kept: the switch is no longer in csrc/sdrfm_b.h)
cd "$GRAFT_REPO_ROOT" || exit 1
CS=stm32f7-rtlsdr_amd/csrc; OUT=gpurun_out/r05_nvt; mkdir -p $OUT
cp $CS/libsdrfm_dev.so $CS/libsdrfm_dev_keep.so
run() {  # label lib R
  cp $CS/$2 $CS/libsdrfm_dev.so
  SDRFM_NO_STREAM=1 SDRFM_FAST_KIND=b SDRFM_FAST_R=$3 timeout 120 python bench.py --dev-library --bit-exact --steps 100 --warmup 10 --no-cpu-baseline --no-steady 2>/dev/null | tail -1 > $OUT/last.json
  python3 - $OUT/last.json "$1" <<'PY'
import json, sys
r = json.load(open(sys.argv[1])); rf = r["roofline"]
print("%-28s %-44s serial %.2f us (sustained %.2f)" % (sys.argv[2], r["config"]["kernel"], rf["kernel_ms_avg"] * 1e3, rf.get("kernel_ms_sustained", 0) * 1e3))
PY
}
{
for rep in 1 2 3; do
  run "R4 taps 12v+20s" libsdrfm_dev_nvt0.so 4
  run "R4 taps 32v" libsdrfm_dev_nvt1.so 4
  run "R12 taps 12v+20s" libsdrfm_dev_nvt0.so 12
  run "R12 taps 32v" libsdrfm_dev_nvt2.so 12
  run "R8 taps 12v+20s" libsdrfm_dev_nvt0.so 8
  run "R8 taps 32v" libsdrfm_dev_nvt2.so 8
done
} | tee $OUT/nvt_$(date +%H%M%S).txt
cp $CS/libsdrfm_dev_keep.so $CS/libsdrfm_dev.so
