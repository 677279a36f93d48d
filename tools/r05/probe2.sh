#!/bin/bash
# tools/r05/probe2.sh — round 5: where a wave's time goes in the sustained regime (300 launches) against a 20-launch burst; does the clock depend on the data?
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe2; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in 1 2 3; do
  for it in 20 300; do
    echo "== phases iters=$it rep=$rep" >> "$OUT/phases.txt"
    QBENCH_STAMPS=1 timeout 120 ./qbench_ph 256 240000 64 5 12 $it fm 2>&1 | grep -E 'phase_cycles|stamps_us|us_per_launch' | sed -E 's/"(kernel|ns|nsamp|T|nslot|runs|first_chunk|checked_streams|worst_at|nonfinite|state_err|batches)":[^,}]*,?//g' >> "$OUT/phases.txt"
    echo "== stamps only iters=$it rep=$rep" >> "$OUT/phases.txt"
    QBENCH_STAMPS=1 timeout 120 ./qbench_st_base 256 240000 64 5 12 $it fm 2>&1 | grep -E 'stamps_us|us_per_launch' | sed -E 's/"(kernel|ns|nsamp|T|nslot|runs|first_chunk|checked_streams|worst_at|nonfinite|state_err|batches)":[^,}]*,?//g' >> "$OUT/phases.txt"
  done
done
for rep in 1 2 3 4; do
  for m in fm const random; do
    for it in 20 300; do
      r=$(QBENCH_GUARD=0 QBENCH_TWO=prio timeout 120 ./qbench_base 256 240000 64 5 12 $it $m 2>&1 | grep -E 'two_streams' | tr '\n' ' ')
      echo "guard-off mode=$m iters=$it rep=$rep $r" >> "$OUT/data_dependence.txt"
    done
  done
done
cat "$OUT/phases.txt" "$OUT/data_dependence.txt"
