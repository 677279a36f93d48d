#!/bin/bash
# quick A/B of fast-kernel tuning knobs on the GPU box: tools/sweep.sh "R=2 AB=5 W=11 MS=4" ...
for cfg in "$@"; do
  unset SDRFM_FAST_KIND SDRFM_WARM_AHEAD SDRFM_FAST_R SDRFM_AUDIO_BATCH SDRFM_WAVES_PER_CU SDRFM_MIN_SUBTILES
  for kv in $cfg; do
    case $kv in
      R=*) export SDRFM_FAST_R=${kv#R=};; K=*) export SDRFM_FAST_KIND=${kv#K=};; WA=*) export SDRFM_WARM_AHEAD=${kv#WA=};; AB=*) export SDRFM_AUDIO_BATCH=${kv#AB=};;
      W=*) export SDRFM_WAVES_PER_CU=${kv#W=};; MS=*) export SDRFM_MIN_SUBTILES=${kv#MS=};;
    esac
  done
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --check 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$cfg', '|', r['config']['kernel'], '| MS/s', r['value'], '| us', r['roofline']['kernel_ms_avg']*1000, '| frac', r['roofline']['frac'], '| parity', r.get('parity_ok'))"
done
