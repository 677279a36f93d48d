#!/bin/bash
# tools/ablate.sh R [W]: time of the fast kernel with phases removed (timing experiments; results are NOT valid audio)
export SDRFM_FAST_R=$1; [ -n "${2:-}" ] && export SDRFM_WAVES_PER_CU=$2
for m in 0 4 6 7; do
  if [ $m = 0 ]; then unset SDRFM_ABLATE; else export SDRFM_ABLATE=$m; fi
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('R=$1 W=${2:-auto} mode=$m', '|', r['config']['kernel'], '| us', round(r['roofline']['kernel_ms_avg']*1000,1))"
done
