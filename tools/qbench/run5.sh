#!/bin/bash
out=gpurun_out/qbench5.jsonl
: > $out
export QBENCH_STAMPS=1
for cfg in "5 12" "5 10" "5 14" "10 8" "10 11" "15 8"; do set -- $cfg; timeout 120 tools/qbench/qbench 256 240000 64 $1 $2 40 fm >> $out 2>&1; done
timeout 120 tools/qbench/qbench 256 240000 16 5 12 40 fm >> $out 2>&1
timeout 120 tools/qbench/qbench 512 240000 64 5 6 40 fm >> $out 2>&1
cat $out
