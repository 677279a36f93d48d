#!/bin/bash
# tools/r05/ab_q.sh <tag> <reps> <variant...> — tools/qbench/qbench_<variant> in turn on ONE box: configs[2] shape, cold inputs rotated over 5 batches = 614 MB;
# per run ten regions of 300 launches back to back, serial and on two streams (the library's overlapped calls).  Reported: the STEADY rate (median of the last
# five regions: the power management's dip between ~1 and ~30 ms after the load starts is over by then) and the first region (from a rested GPU: the dip).
# A variant "name@runs" runs qbench_name with that many runs per stream.
cd "$(dirname "$0")/../qbench" || exit 1
TAG=$1; REPS=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_ab_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    b=${v%%@*}; runs=12; [ "$b" != "$v" ] && runs=${v#*@}
    r=$(QBENCH_REGIONS=10 QBENCH_NB=5 QBENCH_TWO=prio QBENCH_TWO_PREV=1 timeout 120 ./qbench_$b 256 240000 64 5 $runs 300 fm 2>&1 | grep -E 'regions|n_over_tol' | tr '\n' ' ')
    echo "$v rep=$rep $r" >> "$OUT/times.txt"
    sleep 0.2
  done
done
python3 - "$OUT/times.txt" <<'PY'
import re, sys, collections, statistics as st, json
d = collections.defaultdict(lambda: collections.defaultdict(list)); bad = collections.Counter()
for l in open(sys.argv[1]):
    v = l.split()[0]
    ms = re.search(r'"serial_regions_us_per_launch":\[([^\]]*)\]', l); mt = re.search(r'"two_stream_regions_us_per_launch":\[([^\]]*)\]', l)
    m2 = re.search(r'"n_over_tol":(\d+)', l)
    if not (ms and mt and m2) or int(m2.group(1)): bad[v] += 1; continue
    s = [float(x) for x in ms.group(1).split(",")]; t = [float(x) for x in mt.group(1).split(",")]
    d[v]["s_steady"].append(st.median(s[5:])); d[v]["s_first"].append(s[0]); d[v]["t_steady"].append(st.median(t[5:])); d[v]["t_first"].append(t[0])
fr = lambda us: 127795200 / (us * 1e-6) / 8e12
print("variant         serial steady: mean (min .. max) frac | first region |  two streams steady: mean (min .. max) frac | first region")
for v in d:
    a = d[v]
    print("%-14s   %6.2f (%.2f .. %.2f) %.4f |  %6.2f      |   %6.2f (%.2f .. %.2f) %.4f |  %6.2f   %s" % (v, st.mean(a["s_steady"]), min(a["s_steady"]), max(a["s_steady"]), fr(st.mean(a["s_steady"])),
          st.mean(a["s_first"]), st.mean(a["t_steady"]), min(a["t_steady"]), max(a["t_steady"]), fr(st.mean(a["t_steady"])), st.mean(a["t_first"]), "PARITY FAILURES / missing %d" % bad[v] if bad[v] else ""))
for v in bad:
    if v not in d: print(v, "no valid runs", bad[v])
PY
