#!/bin/bash
# tools/r05/stopev2.sh — the library before / after "kernels carry their stream's completion event", 100-call regions, un-profiled and under the kernel tracer (how much of
# the overlapped burst has two kernels resident)
cd "$GRAFT_REPO_ROOT" || exit 1
CS=stm32f7-rtlsdr_amd/csrc; export TMPDIR=/tmp
bash tools/r05/ab_bench.sh libsdrfm_head.so libsdrfm_new.so 4 --steps 100 --warmup 10 --no-steady
OUT=$PWD/gpurun_out/r05_stopev2; rm -rf $OUT; mkdir -p $OUT
for v in head new head new; do
  cp $CS/libsdrfm_$v.so $CS/libsdrfm.so
  rm -rf $OUT/t; rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-steady > $OUT/b.log 2>&1
  python3 tools/overlap_trace_summarize.py "$(find $OUT/t -name '*kernel_trace.csv' | head -1)" $OUT/s.json $v > /dev/null
  python3 -c "
import json; t=json.load(open('$OUT/s.json')); b=t['bursts'][0]; print('$v under the tracer:', {k: v for k, v in b.items() if not isinstance(v, (list, dict))})"
done
cp $CS/libsdrfm_new.so $CS/libsdrfm.so
