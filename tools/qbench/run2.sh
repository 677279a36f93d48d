#!/bin/bash
Q=tools/qbench/qbench
out=gpurun_out/qbench2.jsonl
: > $out
for mode in fm random; do for T in 64 32 16; do timeout 120 $Q 256 240000 $T 10 12 20 $mode >> $out 2>&1; done; done
timeout 120 $Q 64 24000 64 10 3 20 fm >> $out 2>&1
timeout 120 $Q 3 2400000 64 10 300 20 fm >> $out 2>&1
for nslot in 5 10 15; do for runs in 6 8 12 16; do timeout 120 $Q 256 240000 64 $nslot $runs 40 fm >> $out 2>&1; done; done
cat $out
