#!/bin/bash
# tools/r05/soak_head.sh — the randomised soaks at the round's last kernel commit: design Q's guard classes, every parity category, the routing
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_soak_head; mkdir -p $OUT
for seed in 9101 9102 9103 9104; do timeout 400 python tools/fuzz_q.py 300 $seed 2>&1 | tail -2 | cut -c1-600 >> $OUT/fuzz_q.txt; done
timeout 400 python tools/fuzz_parity.py 300 9201 2>&1 | tail -8 | cut -c1-400 > $OUT/fuzz_parity.txt
SDRFM_ROUTE_SOAK=600 timeout 400 python -m pytest tests/test_route_gpu.py -q --timeout 300 -k random_call 2>&1 | tail -2 > $OUT/route_soak.txt
cat $OUT/fuzz_q.txt $OUT/fuzz_parity.txt $OUT/route_soak.txt
