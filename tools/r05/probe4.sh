#!/bin/bash
# tools/r05/probe4.sh — runs cut in quads of four blocks (the wave's step grid starts at its warm-up quad): correctness over geometries, then old vs new
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe4; mkdir -p "$OUT"; export TMPDIR=/tmp
F='s/"(kernel|first_chunk|batches|blocks_per_cu_api)":[^,}]*,?//g'
chk() { echo "== $*" >> "$OUT/check.txt"; env "$@" 2>&1 | grep -E 'from_prev|us_per_launch|error|HIP' | sed -E "$F" >> "$OUT/check.txt"; echo "rc=$?" >> "$OUT/check.txt"; }
# ns nsamp T nslot runs iters mode
chk QBENCH_PREV=1 ./qbench_new 256 240000 64 5 12 3 fm
chk QBENCH_PREV=1 ./qbench_new 256 240000 64 5 12 3 random
chk QBENCH_PREV=1 ./qbench_new 256 240000 16 5 12 3 fm
chk QBENCH_PREV=1 ./qbench_new 64 48000 64 5 7 3 fm
chk QBENCH_PREV=1 ./qbench_new 64 48400 64 5 5 3 random
chk QBENCH_PREV=1 ./qbench_new 8 2400000 64 5 300 3 fm
chk QBENCH_PREV=1 ./qbench_new 512 240000 64 5 6 3 fm
chk QBENCH_PREV=1 ./qbench_new 100 120400 32 5 9 3 random
chk QBENCH_PREV=1 ./qbench_new 256 240000 64 10 8 3 fm
chk QBENCH_PREV=1 QBENCH_D=8 QBENCH_DA=8 ./qbench_new 256 204800 64 4 12 3 fm
chk QBENCH_PREV=1 QBENCH_D=8 QBENCH_DA=8 ./qbench_new 100 102400 16 4 7 3 random
chk QBENCH_PREV=1 QBENCH_D=16 QBENCH_DA=5 ./qbench_new 256 320000 64 8 11 3 fm
chk QBENCH_PREV=1 QBENCH_D=16 QBENCH_DA=5 ./qbench_new 100 160640 16 8 5 3 random
cat "$OUT/check.txt"
for rep in 1 2 3 4 5; do
  for v in old new; do
    for it in 20 300; do
      r=$(QBENCH_TWO=prio timeout 120 ./qbench_$v 256 240000 64 5 12 $it fm 2>&1 | grep -E 'two_streams' | tr '\n' ' ')
      echo "$v iters=$it rep=$rep $r" >> "$OUT/times.txt"
    done
  done
done
for v in old new; do for rep in 1 2 3; do
  QBENCH_STAMPS=1 timeout 120 ./qbench_st_$v 256 240000 64 5 12 20 fm 2>&1 | grep stamps_us | sed "s/^/$v /" >> "$OUT/times.txt"
done; done
cat "$OUT/times.txt"
