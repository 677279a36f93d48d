#!/bin/bash
# tools/pmc_q.sh <tag> [bench args] — SQ / GRBM counter passes for the design Q kernel (k_mfir) under bench.py; one line per counter
TAG=${1:-q}; shift || true
OUT=$PWD/gpurun_out/pmc_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline $*"
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --output-format csv --pmc $grp -d "$OUT/$name" -o pmc -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        if "k_mfir" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print(" ".join("%s=%.4g" % (k, acc[k] / cnt[k]) for k in sorted(acc)))
PY
