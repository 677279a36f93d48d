#!/bin/bash
TAG=${1:-x}; shift || true
OUT=$PWD/gpurun_out/pmc2_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline $*"
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --output-format csv --pmc $grp -d "$OUT/$name" -o pmc -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        if any(k in row["Kernel_Name"] for k in os.environ.get("KFILTER", "k_fast,k_generic").split(",")):
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print(" ".join("%s=%.4g" % (k, acc[k] / cnt[k]) for k in sorted(acc)))
PY
