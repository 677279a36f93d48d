#!/bin/bash
# tools/ab.sh <tag> <reps> <variant...> — same-box A/B of design-Q kernel variants (tools/qbench/qbench_<variant>, built by tools/qbench/build_variants.sh):
# the variants run in turn, <reps> times, on ONE GPU box; configs[2] shape (256 streams x 240 000 samples, 64 taps), cold inputs rotated over 5 batches = 614 MB;
# per run ten regions of 300 launches back to back, serial and on two streams with the previous call's buffer (the library's overlapped calls).  Reported: the
# STEADY rate (median of the last five regions: behind the power management's dip between ~1 and ~30 ms after a load starts) and the first region (the dip).
# Every run also checks the kernel's audio against a float64 restatement of the spec (n_over_tol must be 0).  "name@runs": that many runs per stream.
# Environment: AB_SHAPE="ns nsamp T nslot" (default "256 240000 64 5"), AB_MODE=fm|random|const.  Output: gpurun_out/ab_<tag>/{times.txt,summary.txt}.
cd "$(dirname "$0")/qbench" || exit 1
TAG=$1; REPS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd ../.. && pwd)}
OUT=$ROOT/gpurun_out/ab_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
read -r NS NSAMP T NSLOT <<< "${AB_SHAPE:-256 240000 64 5}"
for rep in $(seq 1 "$REPS"); do
  for v in "$@"; do
    b=${v%%@*}; runs=12; [ "$b" != "$v" ] && runs=${v#*@}
    r=$(QBENCH_REGIONS=10 QBENCH_NB=5 QBENCH_TWO=prio QBENCH_TWO_PREV=1 timeout 120 ./qbench_$b $NS $NSAMP $T $NSLOT $runs 300 ${AB_MODE:-fm} 2>&1 | grep -E 'regions|n_over_tol' | tr '\n' ' ')
    echo "$v rep=$rep $r" >> "$OUT/times.txt"
    sleep 0.2
  done
done
python3 - "$OUT/times.txt" "$NS" "$NSAMP" <<'PY' | tee "$OUT/summary.txt"
import re, sys, collections, statistics as st
d = collections.defaultdict(lambda: collections.defaultdict(list)); bad = collections.Counter(); worst = collections.defaultdict(float)
samples = int(sys.argv[2]) * int(sys.argv[3])
for l in open(sys.argv[1]):
    v = l.split()[0]
    ms = re.search(r'"serial_regions_us_per_launch":\[([^\]]*)\]', l); mt = re.search(r'"two_stream_regions_us_per_launch":\[([^\]]*)\]', l)
    m2 = re.search(r'"n_over_tol":(\d+)', l); mw = re.search(r'"max_scaled_err":([0-9.e+-]+)', l)
    if not (ms and mt and m2) or int(m2.group(1)): bad[v] += 1; continue
    if mw: worst[v] = max(worst[v], float(mw.group(1)))
    s = [float(x) for x in ms.group(1).split(",")]; t = [float(x) for x in mt.group(1).split(",")]
    d[v]["s_steady"].append(st.median(s[5:])); d[v]["s_first"].append(s[0]); d[v]["t_steady"].append(st.median(t[5:])); d[v]["t_first"].append(t[0])
fr = lambda us: samples * 2.0 / (us * 1e-6) / 8e12      # the HBM-READ basis the 70 % target is defined on (2 B per IQ sample)
print("fractions: samples x 2 B / time / 8 TB/s (read basis)")
print("variant         serial steady us: mean (min .. max) frac | first region |  two streams steady us: mean (min .. max) frac | first region | worst err")
for v in d:
    a = d[v]
    print("%-14s   %6.2f (%.2f .. %.2f) %.4f |  %6.2f      |   %6.2f (%.2f .. %.2f) %.4f |  %6.2f  | %.3g %s" % (v, st.mean(a["s_steady"]), min(a["s_steady"]), max(a["s_steady"]), fr(st.mean(a["s_steady"])),
          st.mean(a["s_first"]), st.mean(a["t_steady"]), min(a["t_steady"]), max(a["t_steady"]), fr(st.mean(a["t_steady"])), st.mean(a["t_first"]), worst[v], "PARITY FAILURES / missing %d" % bad[v] if bad[v] else ""))
for v in bad:
    if v not in d: print(v, "no valid runs", bad[v])
PY
