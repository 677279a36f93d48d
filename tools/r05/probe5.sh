#!/bin/bash
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe5; mkdir -p "$OUT"; export TMPDIR=/tmp
F='s/"(kernel|first_chunk|batches|blocks_per_cu_api)":[^,}]*,?//g'
for v in old new; do
  for m in fm random; do
    echo "== $v $m" >> "$OUT/prev.txt"
    QBENCH_PREV=1 ./qbench_$v 256 240000 64 5 12 3 $m 2>&1 | grep -E 'from_prev|us_per_launch' | sed -E "$F" >> "$OUT/prev.txt"
  done
done
cat "$OUT/prev.txt"
for rep in 1 2 3; do for v in old new; do
  QBENCH_STAMPS=1 QBENCH_DUMP=$OUT/dump_${v}_$rep.txt timeout 120 ./qbench_st_$v 256 240000 64 5 12 20 fm 2>&1 | grep -E "stamps_us|two_streams" | sed "s/^/$v /" >> "$OUT/stamps.txt"
done; done
cat "$OUT/stamps.txt"
