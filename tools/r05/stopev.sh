#!/bin/bash
# tools/r05/stopev.sh — overlapped calls' kernels carry their stream's completion event (no marker packets for the joins): tests, the consumer loop, bench A/B
# (csrc/libsdrfm_head.so = the library of the commit before, libsdrfm_new.so = with the change)
cd "$GRAFT_REPO_ROOT" || exit 1
CS=stm32f7-rtlsdr_amd/csrc
timeout 600 python -m pytest tests/test_overlap_gpu.py tests/test_route_gpu.py tests/test_c_frontend_gpu.py tests/test_pcm_sink_gpu.py tests/test_ring_gpu.py -x -q --timeout 300 2>&1 | tail -3
for v in head new head new; do cp $CS/libsdrfm_$v.so $CS/libsdrfm.so; echo "== $v"; timeout 200 python tools/r05/consumer_loop.py 2>&1 | grep flush_previous | sed 's/us per call, ten regions of 300: //'; done
cp $CS/libsdrfm_new.so $CS/libsdrfm.so
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 4 --steps 20 --warmup 5 --no-steady
