#!/bin/bash
# tools/ldsbank/run.sh <set> [args] — on the GPU box: LDS-array cycles per instruction of every pattern of the set
cd "$(dirname "$0")" && export TMPDIR=/tmp
[ -x ./ldsbank ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o ldsbank ldsbank.hip
python3 patterns.py "$@" > patterns.txt
rm -rf /tmp/ldsbank_pmc
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d /tmp/ldsbank_pmc -o pmc -- ./ldsbank patterns.txt > /tmp/ldsbank.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ldsbank_pmc/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "k_pat" in r["Kernel_Name"]:                      # (a small host-to-device copy is a kernel dispatch too)
        d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
labels = open("patterns.labels").read().splitlines()
for (k, v), lab in zip(sorted(d.items()), labels):
    n = v.get("SQ_INSTS_LDS", 0.0) or float("nan")
    print("%-58s %6.2f cycles / instruction   (conflict counter %5.2f)" % (lab, v.get("SQ_LDS_IDX_ACTIVE", 0) / n, v.get("SQ_LDS_BANK_CONFLICT", 0) / n))
PY
