#!/bin/bash
# tools/profile_round.sh <tag> <commit> — the round's rocprofv3 evidence, from HEAD, for every kernel the bench line, BASELINE.md section 3 or DESIGN.md quotes,
# in ONE run on ONE box.  Run on the GPU box from the repo root (gpurun -- 'bash tools/profile_round.sh r06 <commit>'); copy
# gpurun_out/profiles_<tag>/*.{csv,json} into profiles/.  WLS="wbfm spectrum" re-takes a subset.
# Per workload (rocprofv3 runs `python3 bench.py ...` directly, one --pmc group per pass, never mixed with tracing):
#   <tag>_<wl>_kernel_stats.csv           rocprofv3 --kernel-trace --stats (calls one after the other: the tracer's per-kernel duration is the launch's)
#   <tag>_<wl>_bench_under_rocprof.json   the bench JSON line of that run
#   <tag>_<wl>_pmc.json                   per-dispatch means: TCC_EA0_RDREQ by request size (exact fabric read bytes), WRREQ, FETCH_SIZE, WRITE_SIZE,
#                                         SQ_* (VALU busy from SQ_ACTIVE_INST_VALU and SQ_BUSY_CYCLES of ONE pass, instructions per sample and lane, LDS, matrix pipe)
#   <tag>_<wl>_bench.json                 the same workload's un-profiled bench line (the comparison lines of BASELINE.md section 3)
#   <tag>_{fm256,fm512}_overlap_trace.json   kernel trace of the SDRFM_F_OVERLAP calls: interval between calls, kernel duration, residency
#   <tag>_driver_command_*                the driver's exact command under the kernel tracer, its dispatches grouped by burst
set -u
TAG=${1:-r06}; COMMIT=${2:-unknown}
# PLAIN_ONLY=1: only the un-profiled comparison lines (no rocprofv3 passes)
OUT=$PWD/gpurun_out/profiles_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
PROF_ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-steady --no-bit-exact-leg"
run_wl() {   # name, kernel filter, bench args...
  local WL=$1 KF=$2; shift 2
  local W=$OUT/work_$WL; mkdir -p "$W"
  local ARGS="$PROF_ARGS $*"
  if [ -n "${PLAIN_ONLY:-}" ]; then
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bit-exact-leg ${*//--no-overlap/} > "$OUT/${TAG}_${WL}_bench.json" 2> "$W/bench_plain.err"
    return
  fi
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
  python3 tools/profile_pmc_summarize.py "$W" "$OUT" "$TAG" "$WL" "$KF" "$COMMIT"
  # the un-profiled comparison line of the same workload (its own steady series, no CPU baseline)
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bit-exact-leg ${*//--no-overlap/} > "$OUT/${TAG}_${WL}_bench.json" 2> "$W/bench_plain.err"   # (overlapped calls too: the API's fastest way)
}
WLS=${WLS:-"fm256 fm512 fm256_T16 fm256_bitexact mixed10 mixed25 wbfm spectrum fm256_overlap fm512_overlap driver"}
for wl in $WLS; do case $wl in
  fm256)     run_wl fm256 k_mfir --no-overlap ;;
  fm512)     run_wl fm512 k_mfir --streams-per-gpu 512 --no-overlap ;;
  fm256_T16) run_wl fm256_T16 k_mfir --fir-taps 16 --no-overlap ;;
  fm256_D8)  run_wl fm256_D8 k_mfir --fir-decim 8 --no-overlap ;;      # 2.048 MS/s / 8 / 8
  fm256_D16) run_wl fm256_D16 k_mfir --fir-decim 16 --no-overlap ;;    # 3.2 MS/s / 16 / 5
  fm256_bitexact) run_wl fm256_bitexact k_stream --bit-exact --no-overlap ;;
  fm256_pcm) run_wl fm256_pcm k_mfir_pcm --pcm-call pcm --no-overlap ;;   # every call through sdrfm_process_batch_pcm (no audio buffer): the kernel with the sink's chain in it
  mixed10)   run_wl mixed10 k_mix --iq-class mixed:10 --no-overlap ;;   # 10 % noise-only streams: the one-launch kernel (design-B workgroups inside design Q's grid)
  mixed25)   run_wl mixed25 k_mix --iq-class mixed:25 --no-overlap ;;
  wbfm)      run_wl wbfm k_wbfm_ --workload wbfm ;;
  spectrum)  run_wl spectrum k_spectrum --workload spectrum ;;
  fm256_overlap|fm512_overlap|fm256_pcm_overlap)   # SDRFM_F_OVERLAP calls: the kernel trace itself (start / end of every dispatch, queue ids) and what it says
    W=$OUT/work_$wl; mkdir -p "$W"
    EXTRA=""; [ $wl = fm512_overlap ] && EXTRA="--streams-per-gpu 512"; [ $wl = fm256_pcm_overlap ] && EXTRA="--pcm-call pcm"
    rocprofv3 --output-format csv --kernel-trace -d "$W/trace" -o trace -- python3 bench.py $PROF_ARGS $EXTRA > "$W/bench_trace.log" 2>&1
    grep "^{\"metric\"" "$W/bench_trace.log" | tail -1 > "$OUT/${TAG}_${wl}_bench_under_rocprof.json"
    python3 tools/overlap_trace_summarize.py "$(find "$W/trace" -name "*kernel_trace.csv" | head -1)" "$OUT/${TAG}_${wl}_trace.json" "$COMMIT" ;;
  driver)    # the driver's exact command under the kernel tracer, its design-Q dispatches grouped by burst (roofline.frac must be recomputable from profiles/)
    W=$OUT/work_driver; mkdir -p "$W"
    ( cd /tmp && rocprofv3 --output-format csv --kernel-trace --stats -d "$W/trace" -o trace -- python3 $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 > "$W/bench.log" 2>&1 )
    grep "^{\"metric\"" "$W/bench.log" | tail -1 > "$OUT/${TAG}_driver_command_bench_under_rocprof.json"
    find "$W/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_driver_command_kernel_stats.csv" \;
    python3 tools/trace_bursts.py "$(find "$W/trace" -name "*kernel_trace.csv" | head -1)" "$OUT/${TAG}_driver_command_trace.json" "$OUT/${TAG}_driver_command_bench_under_rocprof.json" ;;
esac; done
rm -rf "$OUT"/work_*/trace "$OUT"/work_*/pmc_*/   # (raw traces: tens of MB; the summaries above are what is kept)
ls -la "$OUT"
