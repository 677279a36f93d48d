#!/bin/bash
out=gpurun_out/qbench4.jsonl
: > $out
export QBENCH_STAMPS=1
for a in 0 1 2 3; do for cfg in "5 12" "10 8"; do set -- $cfg; echo "{\"ablate\":$a}" >> $out; timeout 120 tools/qbench/qbench_a$a 256 240000 64 $1 $2 40 fm >> $out 2>&1; done; done
cat $out
