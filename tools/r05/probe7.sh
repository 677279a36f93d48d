#!/bin/bash
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe7; mkdir -p "$OUT"; export TMPDIR=/tmp
F='s/"(kernel|first_chunk|batches|blocks_per_cu_api)":[^,}]*,?//g'
for v in old w4; do
    echo "== $v" >> "$OUT/prev.txt"
    QBENCH_PREV=1 ./qbench_$v 256 240000 64 5 12 3 fm 2>&1 | grep -E 'from_prev|us_per_launch|guard' | sed -E "$F" >> "$OUT/prev.txt"
done
cat "$OUT/prev.txt"
for rep in 1 2 3 4 5 6; do
  for v in old w4; do
    for it in 20 300; do
      r=$(QBENCH_TWO=prio timeout 120 ./qbench_$v 256 240000 64 5 12 $it fm 2>&1 | grep -E 'two_streams' | tr '\n' ' ')
      echo "$v iters=$it rep=$rep $r" >> "$OUT/times.txt"
    done
  done
done
for rep in 1 2; do for v in old w4; do
  QBENCH_STAMPS=1 QBENCH_DUMP=$OUT/dump_${v}_$rep.txt timeout 120 ./qbench_st_$v 256 240000 64 5 12 20 fm 2>&1 | grep -E "stamps_us" | sed "s/^/$v /" >> "$OUT/times.txt"
done; done
sort -s -k1,1 -k2,2 "$OUT/times.txt"
