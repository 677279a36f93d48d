#!/bin/bash
# tools/r05/dist_probe.sh — bench.py's N > 1 code path on one GPU (the rendezvous variables set by hand, world size 1): why are the overlapped calls slow with RCCL initialised?
cd "$GRAFT_REPO_ROOT" || exit 1
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 LOCAL_WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 TORCHELASTIC_RUN_ID=probe
run() { echo "== $*"; local extra=""; [ "$1" = DEV ] && { extra="--dev-library"; shift; }; env "$@" timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-steady --no-cpu-baseline $extra 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); o=r['roofline'].get('overlapped_calls',{}); print('value %.4g ms_per_step %.4f serial %.2f us overlapped %.2f us rccl_world %s' % (r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg']*1e3, o.get('ms_per_call',0)*1e3, r.get('rccl_world')))"; }
run A=1
run GPU_MAX_HW_QUEUES=4
run GPU_MAX_HW_QUEUES=8
run DEV A=1
run DEV SDRFM_Q_NO_ADAPT=1
unset TORCHELASTIC_RUN_ID
run A=plain_no_dist
