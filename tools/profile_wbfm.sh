#!/bin/bash
# tools/profile_wbfm.sh <rNN> — rocprofv3 evidence for `bench.py --workload wbfm` (BASELINE configs[4]); run on the GPU box.
#   <rNN>_wbfm_kernel_stats.csv, <rNN>_wbfm_pmc.json (FETCH_SIZE / WRITE_SIZE raw KiB + SQ counters, per dispatch)
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/profiles_$TAG; W=$OUT/work_wbfm
mkdir -p "$W"; export TMPDIR=/tmp
ARGS="--workload wbfm --steps 30 --warmup 5 --no-cpu-baseline"
rocprofv3 --output-format csv --kernel-trace --stats -d "$W/trace" -o trace -- python3 bench.py $ARGS > "$W/bench_trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --output-format csv --pmc $grp -d "$W/pmc_$name" -o pmc -- python3 bench.py $ARGS > "$W/bench_pmc_$name.log" 2>&1
done
find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_wbfm_kernel_stats.csv" \;
grep "^{\"metric\"" "$W/bench_trace.log" | tail -1 > "$OUT/${TAG}_bench_wbfm_under_rocprof.json"
python3 - "$W" "$OUT/${TAG}_wbfm_pmc.json" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
acc = defaultdict(float); cnt = defaultdict(int)
for fn in glob.glob(os.path.join(sys.argv[1], "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(fn)):
        if "k_wbfm" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
out = {"kernel": "k_wbfm_fused<8,10>", "per_dispatch_mean": {k: acc[k] / cnt[k] for k in sorted(acc)},
       "note": "FETCH_SIZE / WRITE_SIZE are raw KiB (separate passes); the gfx950 x2 read correction of the guide is calibrated for "
               "16-B-per-lane streams and is not applied to this kernel's 2-byte typed loads"}
json.dump(out, open(sys.argv[2], "w"), indent=1); print(json.dumps(out))
PY
head -5 "$OUT/${TAG}_wbfm_kernel_stats.csv"
