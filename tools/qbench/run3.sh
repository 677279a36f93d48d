#!/bin/bash
Q=tools/qbench/qbench
out=gpurun_out/qbench3.jsonl
: > $out
export QBENCH_STAMPS=1
for cfg in "5 12" "5 16" "10 8" "10 12" "15 8"; do set -- $cfg; timeout 120 $Q 256 240000 64 $1 $2 40 fm >> $out 2>&1; done
timeout 120 $Q 512 240000 64 5 12 40 fm >> $out 2>&1
timeout 120 $Q 512 240000 64 5 6 40 fm >> $out 2>&1
cat $out
