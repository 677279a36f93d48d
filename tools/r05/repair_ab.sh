#!/bin/bash
# tools/r05/repair_ab.sh — the repair path's loads all issued ahead of the first conversion (a scheduling barrier): carriers must not change, noise-only input gets cheaper
# (tools/qbench: qbench_prod = the commit before, qbench_rp = with the barrier)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r05/ab_q.sh rp 5 prod rp
cd tools/qbench
for rep in 1 2 3; do for v in prod rp; do echo "$v random: $(timeout 120 ./qbench_$v 256 240000 64 5 12 20 random | grep us_per_launch | sed 's/.*"max_scaled_err":\([^,]*\).*"n_over_tol":\([^,]*\).*"us_per_launch":\([^,]*\).*/err \1 over_tol \2 us \3/')"; done; done
