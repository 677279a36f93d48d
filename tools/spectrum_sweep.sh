#!/bin/bash
# tools/spectrum_sweep.sh — bench.py --workload spectrum over all FFT lengths (run on the GPU box)
for n in 64 128 256 512 1024 2048 4096; do
  python bench.py --workload spectrum --nfft $n --steps 20 --warmup 3 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['config']['kernel'], r['value'], r['roofline']['kernel_ms_avg'], r['roofline']['frac'])"
done
