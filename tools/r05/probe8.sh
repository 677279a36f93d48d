#!/bin/bash
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe8; mkdir -p "$OUT"; export TMPDIR=/tmp
F='s/"(kernel|first_chunk|batches|blocks_per_cu_api)":[^,}]*,?//g'
chk() { echo "== $*" >> "$OUT/check.txt"; env "$@" 2>&1 | grep -E 'from_prev|us_per_launch|error|HIP' | sed -E "$F" >> "$OUT/check.txt"; }
chk QBENCH_PREV=1 ./qbench_g1 256 240000 64 5 12 3 fm
chk QBENCH_PREV=1 ./qbench_g1 256 240000 64 5 12 3 random
chk QBENCH_PREV=1 ./qbench_g1 64 48400 16 5 5 3 random
chk QBENCH_PREV=1 QBENCH_D=8 QBENCH_DA=8 ./qbench_g1 256 204800 64 4 12 3 fm
chk QBENCH_PREV=1 QBENCH_D=16 QBENCH_DA=5 ./qbench_g1 256 320000 64 8 11 3 fm
cat "$OUT/check.txt"
for rep in 1 2 3 4 5 6; do
  for v in old w4 g1s g1; do
    for it in 20 300; do
      r=$(QBENCH_TWO=prio timeout 120 ./qbench_$v 256 240000 64 5 12 $it fm 2>&1 | grep -E 'two_streams' | tr '\n' ' ')
      echo "$v iters=$it rep=$rep $r" >> "$OUT/times.txt"
    done
  done
done
for rep in 1 2; do for v in old g1; do
  QBENCH_STAMPS=1 timeout 120 ./qbench_st_$v 256 240000 64 5 12 20 fm 2>&1 | grep -E "stamps_us" | sed "s/^/$v /" >> "$OUT/times.txt"
done; done
sort -s -k1,1 -k2,2 "$OUT/times.txt"
python3 - "$OUT/times.txt" <<'PY'
import re, sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    m = re.match(r'(\S+) iters=(\d+) rep=\d+ \{"two_streams_us_per_launch":([\d.]+),"one_stream_us_per_launch":([\d.]+)', l)
    if m: d[(m.group(1), int(m.group(2)))].append((float(m.group(3)), float(m.group(4))))
for k in sorted(d):
    import statistics as st
    t = [a for a, b in d[k]]; o = [b for a, b in d[k]]
    print("%-5s iters %3d  serial mean %.2f med %.2f (min %.2f max %.2f) | two-stream mean %.2f med %.2f" % (k[0], k[1], st.mean(o), st.median(o), min(o), max(o), st.mean(t), st.median(t)))
PY
