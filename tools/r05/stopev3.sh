#!/bin/bash
# tools/r05/stopev3.sh — the kernels carry their stream's completion event only for callers that join after every call: tests, the consumer loop, the plain loop under the tracer
# (csrc/libsdrfm_head.so = the library before any of it, libsdrfm_new.so = this one)
cd "$GRAFT_REPO_ROOT" || exit 1
CS=stm32f7-rtlsdr_amd/csrc; export TMPDIR=/tmp
cp $CS/libsdrfm_new.so $CS/libsdrfm.so
timeout 600 python -m pytest tests/test_overlap_gpu.py tests/test_route_gpu.py tests/test_c_frontend_gpu.py tests/test_pcm_sink_gpu.py tests/test_ring_gpu.py -x -q --timeout 300 2>&1 | tail -2
timeout 200 python tools/r05/consumer_loop.py 2>&1 | grep flush_previous | sed 's/us per call, ten regions of 300: //'
OUT=$PWD/gpurun_out/r05_stopev2; rm -rf $OUT; mkdir -p $OUT
for v in head new head new; do
  cp $CS/libsdrfm_$v.so $CS/libsdrfm.so
  rm -rf $OUT/t; rocprofv3 --output-format csv --kernel-trace -d $OUT/t -o trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-steady > $OUT/b.log 2>&1
  python3 tools/overlap_trace_summarize.py "$(find $OUT/t -name '*kernel_trace.csv' | head -1)" $OUT/s.json $v > /dev/null
  python3 -c "
import json; t=json.load(open('$OUT/s.json')); b=t['bursts'][0]; print('$v under the tracer:', {k: v for k, v in b.items() if not isinstance(v, (list, dict))})"
done
cp $CS/libsdrfm_new.so $CS/libsdrfm.so
