#!/bin/bash
# tools/r05/soak_final.sh — a last pass of the randomised soaks at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_soak_final; mkdir -p $OUT
for seed in 9501 9502; do timeout 400 python tools/fuzz_q.py 300 $seed 2>&1 | grep "design-Q soak" | cut -c1-600 >> $OUT/fuzz_q.txt; done
SDRFM_ROUTE_SOAK=600 timeout 400 python -m pytest tests/test_route_gpu.py -q --timeout 300 -k random_call 2>&1 | tail -1 > $OUT/route_soak.txt
timeout 900 python tools/fuzz_parity.py 120 9601 2>&1 | grep "cases" > $OUT/fuzz_parity.txt
cat $OUT/fuzz_q.txt $OUT/route_soak.txt $OUT/fuzz_parity.txt
