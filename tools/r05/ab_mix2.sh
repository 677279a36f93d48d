#!/bin/bash
# tools/r05/ab_mix2.sh — same box, alternating: the library of the commit before (design B's small tile flushes its audio every sub-tile) and the one that flushes every third
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_route_gpu.py -x -q --timeout 200 2>&1 | tail -3
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 3 --steps 100 --warmup 10 --no-steady --iq-class mixed:10
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 3 --steps 100 --warmup 10 --no-steady --iq-class mixed:25
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 1 --steps 100 --warmup 10 --no-steady
