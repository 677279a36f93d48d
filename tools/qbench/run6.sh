#!/bin/bash
out=gpurun_out/qbench6.jsonl
: > $out
export QBENCH_STAMPS=1
for b in qbench qbench_a4 qbench_a7 qbench_nt qbench_nta7; do for cfg in "5 12" "10 8"; do set -- $cfg; echo "{\"bin\":\"$b\"}" >> $out; timeout 120 tools/qbench/$b 256 240000 64 $1 $2 40 fm >> $out 2>&1; done; done
cat $out
