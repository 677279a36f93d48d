#!/bin/bash
# tools/r05/ab_mix.sh — same box, alternating: the library at HEAD (two launches for a mixed batch) and the one with the one-launch kernel (k_mix), on
# all-carrier and mixed batches
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 4 --steps 100 --warmup 10 --no-steady
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 3 --steps 100 --warmup 10 --no-steady --iq-class mixed:10
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 3 --steps 100 --warmup 10 --no-steady --iq-class mixed:25
