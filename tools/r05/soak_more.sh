#!/bin/bash
# tools/r05/soak_more.sh — more of the randomised soaks at HEAD (other seeds)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_soak_more; mkdir -p $OUT
for seed in 9301 9302 9303 9304 9305; do timeout 400 python tools/fuzz_q.py 300 $seed 2>&1 | grep "design-Q soak" | cut -c1-600 >> $OUT/fuzz_q.txt; done
SDRFM_ROUTE_SOAK=1200 timeout 600 python -m pytest tests/test_route_gpu.py -q --timeout 300 -k random_call 2>&1 | tail -1 > $OUT/route_soak.txt
timeout 400 python tools/fuzz_stream.py 240 9401 2>&1 | tail -3 | cut -c1-400 > $OUT/fuzz_stream.txt
cat $OUT/fuzz_q.txt $OUT/route_soak.txt $OUT/fuzz_stream.txt
