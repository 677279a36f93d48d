#!/bin/bash
# tools/r05/probe3.sh — round 5: what the table loads cost (guard off in both arms, so that the synthetic tables of -DSDRFM_Q_ABLATE=4 do not send every lane
# to the repair path), and runs per stream 12..15 at HEAD
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe3; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in 1 2 3 4 5; do
  for v in base a4; do
    r=$(QBENCH_GUARD=0 QBENCH_TWO=prio timeout 120 ./qbench_$v 256 240000 64 5 12 20 fm 2>&1 | grep -E 'two_streams' | tr '\n' ' ')
    echo "guard-off $v iters=20 rep=$rep $r" >> "$OUT/tables.txt"
  done
done
for v in base a4; do for rep in 1 2 3; do
  QBENCH_GUARD=0 QBENCH_STAMPS=1 timeout 120 ./qbench_st_$v 256 240000 64 5 12 20 fm 2>&1 | grep stamps_us | sed "s/^/$v /" >> "$OUT/tables.txt"
done; done
for rep in 1 2 3; do
  for runs in 12 13 14 15; do
    for it in 20 300; do
      r=$(QBENCH_TWO=prio timeout 120 ./qbench_base 256 240000 64 5 $runs $it fm 2>&1 | grep -E 'two_streams' | tr '\n' ' ')
      echo "runs=$runs iters=$it rep=$rep $r" >> "$OUT/runs.txt"
    done
  done
done
cat "$OUT/tables.txt" "$OUT/runs.txt"
