#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05_session_a; mkdir -p $OUT
timeout 200 python -m pytest tests/test_route_gpu.py tests/test_q_guard_gpu.py tests/test_overlap_gpu.py -x -q --timeout 60 2>&1 | tail -4
echo "== mixed with any-order launches"
bash tools/r05/mixed.sh 2>&1 | head -11
echo "== trace"
bash tools/r05/mixed_trace.sh mixed:10 2>&1 | head -30
echo "== a call of K times the bytes (stream-major ring): qbench, serial, regions of 100 launches"
cd tools/qbench
for k in 1 2 4; do
  n=$((240000 * k))
  QBENCH_REGIONS=8 QBENCH_NB=$((k == 1 ? 5 : 3)) QBENCH_TWO=prio QBENCH_TWO_PREV=1 timeout 200 ./qbench_prod 256 $n 64 5 12 100 fm 2>&1 | grep -E "regions|n_over_tol" | sed "s/^/K=$k /" | cut -c1-400
done
