#!/bin/bash
# tools/profile_round5.sh <tag> <commit> — the round's rocprofv3 evidence (round 4's script; bench.py without its steady-state series; the driver's exact command traced and grouped by burst), from HEAD, for every kernel the bench line or DESIGN.md
# quotes.  Run on the GPU box from the repo root; copy gpurun_out/profiles_<tag>/*.{csv,json} into profiles/.
# Per workload (rocprofv3 runs `python3 bench.py ...` directly, one --pmc group per pass, never mixed with tracing):
#   <tag>_<wl>_kernel_stats.csv       rocprofv3 --kernel-trace --stats
#   <tag>_<wl>_bench_under_rocprof.json   the bench JSON line of that run
#   <tag>_<wl>_pmc.json               per-dispatch means: TCC_EA0_RDREQ by request size (exact fabric read bytes), WRREQ, FETCH_SIZE,
#                                     WRITE_SIZE, GRBM_GUI_ACTIVE, SQ_* (VALU busy etc.)
set -u
TAG=${1:-r05}; COMMIT=${2:-unknown}
OUT=$PWD/gpurun_out/profiles_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
run_wl() {   # name, kernel filter, bench args...
  local WL=$1 KF=$2; shift 2
  local W=$OUT/work_$WL; mkdir -p "$W"
  local ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-steady $*"
  rocprofv3 --output-format csv --kernel-trace --stats -d "$W/trace" -o trace -- python3 bench.py $ARGS > "$W/bench_trace.log" 2>&1
  find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_${WL}_kernel_stats.csv" \;
  grep "^{\"metric\"" "$W/bench_trace.log" | tail -1 > "$OUT/${TAG}_${WL}_bench_under_rocprof.json"
  for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
             "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8"; do
    name=$(echo $grp | tr ' ' '_' | cut -c1-30)
    rocprofv3 --output-format csv --pmc $grp -d "$W/pmc_$name" -o pmc -- python3 bench.py $ARGS > "$W/pmc_$name.log" 2>&1
  done
  python3 tools/profile_round3_summarize.py "$W" "$OUT" "$TAG" "$WL" "$KF" "$COMMIT"
}
WLS=${WLS:-"fm256 fm512 fm256_T16 fm256_D8 fm256_D16 fm256_overlap fm512_overlap driver"}      # WLS="wbfm" re-takes one workload only
for wl in $WLS; do case $wl in
  # (--no-overlap: every call after the previous one, so that the tracer's per-kernel duration is the launch's duration; the overlapped
  #  calls the bench line's `value` is measured on are traced separately below)
  fm256)     run_wl fm256 k_mfir --no-overlap ;;
  fm512)     run_wl fm512 k_mfir --streams-per-gpu 512 --no-overlap ;;
  fm256_T16) run_wl fm256_T16 k_mfir --fir-taps 16 --no-overlap ;;
  fm256_D8)  run_wl fm256_D8 k_mfir --fir-decim 8 --no-overlap ;;      # 2.048 MS/s / 8 / 8 (round 4: design Q at the other front-end rates)
  fm256_D16) run_wl fm256_D16 k_mfir --fir-decim 16 --no-overlap ;;    # 3.2 MS/s / 16 / 5
  fm256_bitexact) run_wl fm256_bitexact k_stream --bit-exact --no-overlap ;;
  mixed10)   run_wl mixed10 k_mix --iq-class mixed:10 --no-overlap ;;   # 10 % noise-only streams: the one-launch kernel (design-B workgroups inside design Q's grid)
  fm256_overlap|fm512_overlap)   # SDRFM_F_OVERLAP calls: the kernel trace itself (start / end of every dispatch, queue ids) and what it says
    W=$OUT/work_$wl; mkdir -p "$W"
    EXTRA=""; [ $wl = fm512_overlap ] && EXTRA="--streams-per-gpu 512"
    rocprofv3 --output-format csv --kernel-trace -d "$W/trace" -o trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-steady $EXTRA > "$W/bench_trace.log" 2>&1
    grep "^{\"metric\"" "$W/bench_trace.log" | tail -1 > "$OUT/${TAG}_${wl}_bench_under_rocprof.json"
    python3 tools/overlap_trace_summarize.py "$(find "$W/trace" -name "*kernel_trace.csv" | head -1)" "$OUT/${TAG}_${wl}_trace.json" "$COMMIT" ;;
  driver)    # the driver's exact command under the kernel tracer, its design-Q dispatches grouped by burst (roofline.frac must be recomputable from profiles/)
    W=$OUT/work_driver; mkdir -p "$W"
    ( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$W/trace" -o trace -- python3 $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 > "$W/bench.log" 2>&1 )
    grep "^{\"metric\"" "$W/bench.log" | tail -1 > "$OUT/${TAG}_driver_command_bench_under_rocprof.json"
    find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_driver_command_kernel_stats.csv" \;
    python3 tools/r05/trace_bursts.py "$(find "$W/trace" -name "*kernel_trace.csv" | head -1)" "$OUT/${TAG}_driver_command_trace.json" "$OUT/${TAG}_driver_command_bench_under_rocprof.json" ;;
  wbfm)      run_wl wbfm k_wbfm_ --workload wbfm ;;
  spectrum)  run_wl spectrum k_spectrum --workload spectrum ;;
esac; done
ls -la "$OUT"
