#!/bin/bash
# tools/r05/probe1.sh — round 5, first measurement pass (GPU box): what the table loads, the audio stores and the kernel boundary cost design Q,
# with the ablation switches the kernel already has; alternating runs on one box.  Writes gpurun_out/r05_probe1/*.
cd "$(dirname "$0")/../qbench" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_probe1; mkdir -p "$OUT"; export TMPDIR=/tmp
V="base a4 a128 a256"
for rep in 1 2 3 4 5; do
  for v in $V; do
    for it in 20 300; do
      r=$(QBENCH_TWO=prio timeout 120 ./qbench_$v 256 240000 64 5 12 $it fm 2>&1 | grep -E 'two_streams|us_per_launch' | tr '\n' ' ')
      echo "$v iters=$it rep=$rep $r" | sed -E 's/"(kernel|ns|nsamp|T|nslot|runs|mode|first_chunk|checked_streams|worst_at|nonfinite|state_err|batches)":[^,}]*,?//g' >> "$OUT/times.txt"
    done
  done
done
# per-wave stamps: first bytes / ends with and without the table loads
for v in base a4; do
  for rep in 1 2 3; do
    QBENCH_STAMPS=1 timeout 120 ./qbench_st_$v 256 240000 64 5 12 20 fm 2>&1 | grep stamps_us >> "$OUT/stamps_$v.txt"
  done
done
# kernel trace: duration and gap to the next launch, serial launches
for v in base a128 a256; do
  rm -rf /tmp/tr_$v
  rocprofv3 --output-format csv --kernel-trace -d /tmp/tr_$v -o t -- ./qbench_$v 256 240000 64 5 12 40 fm > /dev/null 2>&1
  python3 - /tmp/tr_$v $v >> "$OUT/trace.txt" <<'PY'
import csv, glob, os, sys
import numpy as np
rows = []
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_mfir" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
s = np.array([a for a, b in rows]); e = np.array([b for a, b in rows])
dur = (e - s) / 1e3; gap = (s[1:] - e[:-1]) / 1e3
g = gap[gap < 20]
print(sys.argv[2], "launches", len(dur), "dur us mean %.2f med %.2f min %.2f | gap (back-to-back) med %.2f mean %.2f n %d" % (dur[6:].mean(), np.median(dur[6:]), dur[6:].min(), np.median(g), g.mean(), len(g)))
PY
done
# LDS access patterns of k_mfir (tools/ldsbank)
cd ../ldsbank && bash run.sh mfir > "$OUT/ldsbank_mfir.txt" 2>&1
cat "$OUT/times.txt" | tail -50; cat "$OUT"/stamps_*.txt "$OUT/trace.txt" "$OUT/ldsbank_mfir.txt"
