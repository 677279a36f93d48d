#!/bin/bash
# tools/profile_round.sh <rNN> — run on the GPU box from the repo root.
# Produces the rocprofv3 evidence the bench line refers to, under gpurun_out/profiles_<rNN>/ (copy into profiles/):
#   <rNN>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python3 bench.py`
#   <rNN>_pmc_summary.json      per-dispatch means of the SQ / LDS counters (separate --pmc passes)
#   traffic_<rNN>.json          HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (separate passes, gfx950 correction applied)
set -u
TAG=${1:-r01}; shift || true
OUT=$PWD/gpurun_out/profiles_$TAG; W=$OUT/work
mkdir -p "$W"; export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --output-format csv --kernel-trace --stats -d "$W/trace" -o trace -- python3 bench.py $ARGS > "$W/bench_trace.log" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --output-format csv --pmc $grp -d "$W/pmc_$name" -o pmc -- python3 bench.py $ARGS > "$W/bench_pmc_$name.log" 2>&1
done
cp "$W/trace/trace_kernel_stats.csv" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
grep "^{\"metric\"" "$W/bench_trace.log" | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
python3 tools/profile_summarize.py "$W" "$OUT" "$TAG"
