#!/bin/bash
# Usage (on the GPU box, from repo root): tools/prof.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/ and prints compact summaries.
set -u
TAG=${1:-r1}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/trace" -o trace -- python3 bench.py $ARGS > "$OUT/bench_trace.log" 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $grp -d "$OUT/pmc_$name" -o pmc -- python3 bench.py $ARGS > "$OUT/bench_pmc_$name.log" 2>&1
done
python3 tools/prof_summary.py "$OUT"
