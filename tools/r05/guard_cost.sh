#!/bin/bash
# tools/r05/guard_cost.sh — VERDICT r04 item 4: what design Q's conditioning guard costs on carriers (it never fires there): the product kernel against the same
# source with the guard compiled out (-DSDRFM_Q_ABLATE=512), ten alternating runs per cell on ONE box; configs[2] shape, cold inputs rotated over 5 batches.
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_guard_cost; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in $(seq 1 10); do
  for v in ${VARIANTS:-prod noguard}; do
    a=$(QBENCH_REGIONS=10 QBENCH_NB=5 QBENCH_TWO=prio QBENCH_TWO_PREV=1 timeout 120 ./qbench_$v 256 240000 64 5 12 300 fm 2>&1 | grep -E 'regions' | tr '\n' ' ')
    sleep 0.3
    b=$(QBENCH_REGIONS=2 QBENCH_NB=5 QBENCH_TWO=prio QBENCH_TWO_PREV=1 timeout 120 ./qbench_$v 256 240000 64 5 12 20 fm 2>&1 | grep -E 'regions' | tr '\n' ' ')
    echo "$v rep=$rep L300 $a L20 $b" >> "$OUT/raw.txt"
    sleep 0.3
  done
done
python3 - "$OUT/raw.txt" <<'PY'
import re, sys, statistics as st, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    v = l.split()[0]
    l300, l20 = l.split(" L20 ")
    def arr(s, key):
        m = re.search(r'"%s":\[([^\]]*)\]' % key, s); return [float(x) for x in m.group(1).split(",")]
    s3, t3, s2, t2 = arr(l300, "serial_regions_us_per_launch"), arr(l300, "two_stream_regions_us_per_launch"), arr(l20, "serial_regions_us_per_launch"), arr(l20, "two_stream_regions_us_per_launch")
    d[v]["serial, steady (median of regions 6-10 of 300 launches)"].append(st.median(s3[5:]))
    d[v]["serial, first 300 launches from rest"].append(s3[0])
    d[v]["serial, first 20 launches from rest"].append(s2[0])
    d[v]["two streams, steady"].append(st.median(t3[5:]))
    d[v]["two streams, first 300 launches"].append(t3[0])
    d[v]["two streams, first 20 launches"].append(t2[0])
print("%-62s %-28s %-28s %s" % ("us per launch: mean +- standard deviation (min .. max), n = 10", "product (guard never fires)", "guard compiled out", "difference"))
for key in d["prod"]:
    a, b = d["prod"][key], d["noguard"][key]
    f = lambda x: "%.2f +- %.2f (%.2f .. %.2f)" % (st.mean(x), st.pstdev(x), min(x), max(x))
    print("%-62s %-28s %-28s %+.2f us = %+.1f %%" % (key, f(a), f(b), st.mean(a) - st.mean(b), 100 * (st.mean(a) - st.mean(b)) / st.mean(b)))
PY
