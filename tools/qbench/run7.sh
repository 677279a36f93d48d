#!/bin/bash
out=gpurun_out/qbench7.jsonl
: > $out
export QBENCH_STAMPS=1
for nb in 1 3; do for cfg in "5 12" "10 8" "15 8" "5 8"; do set -- $cfg; echo "{\"nb\":$nb}" >> $out; QBENCH_NB=$nb timeout 120 tools/qbench/qbench 256 240000 64 $1 $2 40 fm >> $out 2>&1; done; done
cat $out
