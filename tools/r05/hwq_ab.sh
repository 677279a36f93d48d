#!/bin/bash
# tools/r05/hwq_ab.sh — the plain N = 1 driver-style bench line with the runtime's default 4 hardware queues and with 8 (what bench.py now asks for), alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3 4 5; do for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); rf=r['roofline']; o=rf['overlapped_calls']; print('queues $q  value %.4g | serial frac %.4f steady %s | overlapped frac %.4f sustained %.4f steady %s' % (r['value'], rf['frac'], rf.get('frac_steady'), o['frac'], o['frac_sustained'], o.get('frac_steady')))"
done; done
