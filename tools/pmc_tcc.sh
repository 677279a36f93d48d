#!/bin/bash
# tools/pmc_tcc.sh <tag> [bench args] — exact fabric read bytes of the dominant kernel: TCC_EA0_RDREQ split by request size
# (32 / 64 / 128 B) and the L2 hit / miss counts, one rocprofv3 --pmc pass per group (TCC has 4 slots).
TAG=${1:-x}; shift || true
OUT=$PWD/gpurun_out/tcc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline $*"
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum TCP_TCC_READ_REQ_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --output-format csv --pmc $grp -d "$OUT/$name" -o pmc -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, os, json
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int); kn = None
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        if any(k in row["Kernel_Name"] for k in ("k_fast", "k_stream", "k_wbfm_fused", "k_spectrum")):
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1; kn = row["Kernel_Name"]
m = {k: acc[k] / cnt[k] for k in acc}
n32, n64, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0)
m["read_bytes_by_request_size"] = 32 * n32 + 64 * n64 + 128 * n128
m["kernel"] = kn; m["tag"] = sys.argv[2]
print(json.dumps(m))
PY
