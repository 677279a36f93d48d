#!/usr/bin/env python3
"""bench.py — IQ MSamples/s through FIR + FM-demod + resample on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (sdrfm_process_batch on device-resident buffers: one fused kernel launch) over one
batch of synthetic IQ:
  N = 1   BASELINE configs[2] — 256 concurrent 2.4 MS/s streams x 0.1 s (480 000 B each), 64-tap FIR /10, FM discriminator,
          32-tap /5 audio resampler;
  N > 1   BASELINE configs[3]'s share per GPU — 512 streams each (4096 streams on 8 GPUs); streams are independent, so every
          rank owns its own streams on its own GPU: no data-path collective, weak scaling.
`python3 bench.py --gpus N` works as typed: with N > 1 and no torch.distributed environment it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process (before anything touches the GPU),
forwards its JSON line and exits with its return code.

Cold inputs: the timed loop rotates over enough distinct input batches (>= 3, > 256 MiB in total, i.e. more than the
Infinity Cache holds) that every step reads its bytes from HBM; that is `value`.  The figure with ONE resident 123 MB
batch (which the 256 MiB L3 can keep) is reported beside it as `resident_input`, never as `value`.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).  The GPU legs never touch oracle/; only the
`cpu_baseline` leg (rank 0, N=1) times the scalar-C oracle on a bounded sample of the same workload.
"""
import argparse
import glob
import importlib
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The HIP runtime maps a process's streams onto a small pool of hardware queues (4 by default).  With RCCL initialised in the process its streams take some of them,
# and the library's two internal streams of the overlapped calls then share queues with the launch stream: every call waits for the one before it across queues —
# 49 us per call instead of 23 on one GPU with world size 1 (profiles/r05_q_experiments.txt item 16).  Eight queues restore it.  Set before
# anything initialises the runtime; a caller's own setting is respected.  (INTEGRATION.md says the same to hosts that run RCCL beside the library.)
if "WORLD_SIZE" in os.environ or "TORCHELASTIC_RUN_ID" in os.environ:   # (a rank of a distributed run; the plain N = 1 run keeps the runtime's default: 4 and 8 queues measure the same there)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
REST_S = float(os.environ.get("BENCH_REST_S", "0.05"))   # idle time before every secondary timed region (see timed())
L3_BYTES = 256 << 20            # Infinity Cache (same guide): the rotated inputs must exceed it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams-per-gpu", type=int, default=0, help="0 = 256 at one GPU (configs[2]), 512 at several (configs[3] share)")
    ap.add_argument("--seconds", type=float, default=0.1, help="capture length per stream per step")
    ap.add_argument("--fir-taps", type=int, default=64)
    ap.add_argument("--fir-decim", type=int, default=10, choices=[8, 10, 16], help="fm workload: the front end's rate — 10 = 2.4 MS/s / 10 / 5 (BASELINE, "
                    "the headline), 8 = 2.048 MS/s / 8 / 8, 16 = 3.2 MS/s / 16 / 5 (the other rates RTLSDR_set_sample_rate accepts); comparison figures")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic streams generated on the host per rank; the other "
                    "rows and the other rotated batches are byte-rotations of them made on the device (0 = generate every row)")
    ap.add_argument("--batches", type=int, default=0, help="input batches the timed loop rotates over (0 = as many as exceed the L3, >= 3)")
    ap.add_argument("--iq-class", default="fm", help="fm workload: the synthetic input class — fm = the FM test signal of "
                    "SURVEY.md 8d (default, the headline); random = uniform random bytes, its worst-case class (noise only: the matrix-pipe kernel's "
                    "conditioning guard sends about one lane in ten to the repair path); mixed:P = P per cent of the streams (evenly spread) hold uniform "
                    "random bytes, the others the FM test signal (dongles that are not tuned to a station: the library routes them per stream); "
                    "comparison figures, labelled as such")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-consumer-leg", action="store_true", help="skip the consumer-loop leg (sdrfm_process_batch_pcm, steady state; fm workload)")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state series (ten regions of 300 calls; fm workload)")
    ap.add_argument("--no-bit-exact-leg", action="store_true", help="skip the SDRFM_CFG_BIT_EXACT comparison leg of the default line (bit_exact_kernel)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--check", action="store_true", help="also verify a few streams against the oracle (not timed)")
    ap.add_argument("--workload", choices=["fm", "wbfm", "spectrum"], default="fm",
                    help="fm = BASELINE configs[2]/[3] (default, the headline); wbfm = configs[4] channelizer path, 128 streams/GPU; "
                         "spectrum = FFT view (SURVEY 8f-3) of the configs[2] buffers")
    ap.add_argument("--nfft", type=int, default=1024, help="spectrum workload: FFT length")
    ap.add_argument("--dev-library", action="store_true", help="fm / spectrum workloads: load csrc/libsdrfm_dev.so (honours the SDRFM_* development "
                    "knobs, e.g. SDRFM_NO_STREAM=1 for design B); the reported line then says so and is not a product figure")
    ap.add_argument("--bit-exact", action="store_true", help="fm workload: SDRFM_CFG_BIT_EXACT handle — the fmaf-chain kernels only (design S "
                    "instead of the matrix-pipe design Q); a comparison figure, labelled as such")
    ap.add_argument("--pcm-call", choices=["none", "pcm", "both"], default="none",
                    help="fm workload: make EVERY call of the run through sdrfm_process_batch_pcm (the board's sink format out of the demodulator's own launch: "
                         "'pcm' without an audio buffer, 'both' with one) — for profiling that kernel; the default line measures the plain calls and reports the "
                         "PCM calls in its consumer_loop block")
    ap.add_argument("--no-overlap", action="store_true", help="fm workload: make the timed calls one after the other (without SDRFM_F_OVERLAP); "
                    "by default consecutive calls may overlap on the device (two audio buffers in turn) and `value` is that throughput, "
                    "while `roofline` is always taken from calls made one after the other — the duration rocprofv3 reports per kernel")
    ap.add_argument("--end-to-end", action="store_true",
                    help="fm workload: every step also scatters the IQ batch from rank 0 to all ranks and gathers the audio back "
                         "over RCCL (SURVEY 8e C1/C2); reported separately from the compute-only default")
    args = ap.parse_args()
    if not (args.iq_class in ("fm", "random") or (args.iq_class.startswith("mixed:") and args.iq_class[6:].isdigit() and 0 < int(args.iq_class[6:]) < 100)):
        ap.error("--iq-class: fm, random or mixed:P (P per cent, 1..99)")
    return args


def self_launch(args):
    """`bench.py --gpus N` typed directly: run the N ranks as a child job and relay its one JSON line.  Called before torch (or
    anything else) has touched the GPU; this process never does."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line:
        print(line, flush=True)
    sys.exit(proc.returncode if proc.returncode else (0 if line else 1))


def latest_traffic(kernel_name, alg_bytes=None):
    """HBM bytes per launch from the newest committed PMC summary (profiles/traffic_r*.json, profiles/r*_pmc.json) taken for this
    kernel AND this workload size (same algorithmic bytes per launch); None when no committed profile matches."""
    best = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc.json")) + glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")),
                   key=lambda fn: (re.match(r"(?:traffic_)?(r\d+[a-z]?)", os.path.basename(fn)).group(1), os.path.basename(fn).startswith("traffic_")))
    for fn in files:
        try:
            with open(fn) as f:
                t = json.load(f)
            if t.get("kernel_name") != kernel_name or "hbm_bytes_per_launch" not in t:
                continue
            if str(t.get("workload", "")).endswith("_pcm") != PCM_CALL["on"]:   # (the counters of --pcm-call runs belong to those runs only)
                continue
            ab = t.get("algorithmic_bytes_per_launch")
            if alg_bytes is not None and ab and abs(ab - alg_bytes) > 1e-3 * alg_bytes:
                continue
            best = dict(t, file=os.path.basename(fn))
        except Exception:
            pass
    return best


def latest_pmc_derived(kernel_name):
    """Derived figures (VALU-issue busy fraction, clock, ...) of the newest committed counter pass of this kernel (profiles/r*_pmc.json)."""
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*_pmc.json")), key=lambda f: re.match(r"(r\d+[a-z]?)", os.path.basename(f)).group(1)):
        try:
            with open(fn) as f:
                t = json.load(f)
            if t.get("kernel_name") == kernel_name and t.get("derived"):
                best = dict(t["derived"], file=os.path.basename(fn), commit=t.get("commit"),
                            lds=t.get("counters_per_dispatch_mean", {}))
        except Exception:
            pass
    return best


def bound_block(kind, kname, ms, alg, traffic):
    """`roofline` of a kernel that is NOT bound by the HBM: the HBM figure stays (BASELINE's metric is quoted against it) but `bound` names the pipe
    that limits the kernel, with the busy fraction of that pipe from the newest committed counter pass — how close the kernel is to ITS roofline."""
    d = latest_pmc_derived(kname) or {}
    blk = {"bound": kind, "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "frac_note": "achieved / peak are the HBM figures (algorithmic bytes / launch duration against 8 TB/s); the kernel is bound by the %s pipe, see below" % ("vector" if kind == "valu" else "LDS"),
           "traffic": traffic, "kernel_ms_avg": round(ms, 4), "algorithmic_bytes_per_launch": alg}
    if d:
        blk["valu_issue_busy_fraction"] = round(d.get("valu_issue_busy_fraction", 0.0), 3)
        blk["valu_insts_per_iq_sample_per_lane"] = d.get("valu_insts_per_iq_sample_per_lane")
        blk["wave_cycles_parked_in_waitcnt_fraction"] = round(d.get("wave_cycles_parked_in_waitcnt_fraction", 0.0), 3)
        c = d.get("lds", {})
        if c.get("SQ_LDS_IDX_ACTIVE") and c.get("SQ_BUSY_CYCLES"):
            blk["lds_active_cycles_per_cu_over_kernel_cycles"] = round(c["SQ_LDS_IDX_ACTIVE"] / 256.0 / (c["SQ_BUSY_CYCLES"] / 32.0), 3)
        blk["instruction_floor_ms"] = round(ms * blk["valu_issue_busy_fraction"], 4) if kind == "valu" else None
        if kind == "valu":
            # the kernel's OWN roofline: the vector pipe issues one wave-instruction per SIMD and issue cycle; valu_frac = issue cycles the kernel's vector
            # instructions need (counted: SQ_ACTIVE_INST_VALU) / issue cycles the launch had (SQ_BUSY_CYCLES x SIMDs), both from ONE counter pass
            blk["valu_frac"] = blk["valu_issue_busy_fraction"]
            blk["valu_frac_note"] = ("fraction of the vector pipe's issue cycles this kernel's instructions occupy (4 SQ_ACTIVE_INST_VALU / 1024 SIMDs / kernel cycles, one counter "
                                     "pass): the roofline that binds it; the HBM fraction above is what BASELINE's metric is quoted against")
        blk["pipe_figures_source"] = "profiles/%s (separate rocprofv3 --pmc passes at commit %s)" % (d["file"], d.get("commit", "?"))
    return blk


def cpu_baseline(pkg, h, g, iq_host, seconds, threads):
    """Scalar-C oracle on the GPU box's host cores, bounded sample of the same workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.oracle import Oracle
    n_streams, nbytes = iq_host.shape
    nsamp = nbytes // 2

    def run(nthreads, budget):
        orcs = [Oracle(h, g) for _ in range(nthreads)]
        done, t0 = 0, time.perf_counter()

        def work(i):
            k = 0
            while time.perf_counter() - t0 < budget:
                orcs[i].reset()
                orcs[i].process(iq_host[(i + k * nthreads) % n_streams])
                k += 1
            return k
        if nthreads == 1:
            done = work(0)
        else:
            with ThreadPoolExecutor(nthreads) as ex:
                done = sum(ex.map(work, range(nthreads)))
        dt = time.perf_counter() - t0
        return done * nsamp / dt / 1e6, done

    v1, n1 = run(1, seconds)
    cpu_model, compiler = "unknown", "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name"))
    except Exception:
        pass
    try:
        compiler = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0].strip()
    except Exception:
        pass
    out = {"value": round(v1, 3), "unit": "MSamples/s", "cores": 1, "kind": "port", "cpu_model": cpu_model,
           "host_cores": os.cpu_count() or 1, "compiler": compiler, "compiler_flags": "-O2 -std=c99 -ffp-contract=off -mfma (oracle/Makefile)",
           "sample": "%d stream-chunks of the workload (%.1f s x 2.4 MS/s each), scalar-C oracle, 1 thread, %.0f s" % (n1, nsamp / 2.4e6, seconds)}
    if threads > 1:
        vn, nn = run(threads, max(3.0, seconds / 2))
        out["all_cores"] = {"value": round(vn, 3), "cores": threads, "stream_chunks": nn}
    return out


def noisy_rows(ns, percent):
    """the streams of a mixed:P batch that hold noise: P per cent of them, evenly spread"""
    return [s for s in range(ns) if ((s + 1) * percent) // 100 > (s * percent) // 100]


def make_batches(torch, pkg, stream, ns, nsamp, fs, rank, distinct, nbatches, mode="fm"):
    """nbatches device-resident input batches [ns, 2*nsamp] u8 that are CONSECUTIVE pieces of every stream: batch b holds samples
    [b nsamp, (b + 1) nsamp) of a capture nbatches x nsamp samples long, as the buffers of a front end that hands over a running
    capture are (the reference re-arms the same pipe again and again: usbh_rtlsdr.c:1058-1101).  `distinct` captures come from the
    host generator (own PRNG stream per (rank, row)); every other row is one of them rotated by a different whole number of I/Q pairs
    over its whole length on the device: the same signal statistics at different bytes, so no two rows (or batches) share a cache
    line.  A stream's signal is therefore continuous from call to call, but for the one place per nbatches calls where the rotation
    wraps and the one where the timed loop starts the capture over (rounds 1 - 3 drew every batch independently: two phase jumps per
    stream and call, which a demodulator fed by a running capture never sees)."""
    import numpy as np
    distinct = ns if distinct <= 0 else min(distinct, ns)
    total = nbatches * nsamp
    t0 = time.perf_counter()
    mixed_percent = int(mode.split(":")[1]) if mode.startswith("mixed:") else 0
    if mixed_percent:
        mode = "fm"
    base_host = pkg.make_iq(distinct, total, mode=mode, fs=fs, first_id=rank * ns)
    t_gen = time.perf_counter() - t0
    with torch.cuda.stream(stream):
        base = torch.from_numpy(base_host).cuda()
        batches = [torch.empty((ns, 2 * nsamp), dtype=torch.uint8, device="cuda") for _ in range(nbatches)]
        for r0 in range(0, ns, distinct):
            k = r0 // distinct
            rows = base if k == 0 else torch.roll(base, shifts=2 * (7919 * k % total), dims=1)
            n = min(distinct, ns - r0)
            for b in range(nbatches):
                batches[b][r0:r0 + n] = rows[:n, 2 * b * nsamp:2 * (b + 1) * nsamp]
            del rows
        del base
        if mixed_percent:
            rows = torch.tensor(noisy_rows(ns, mixed_percent), dtype=torch.long, device="cuda")
            gen = torch.Generator(device="cuda")
            gen.manual_seed(1234 + rank)
            for b in range(nbatches):
                batches[b][rows] = torch.randint(0, 256, (rows.numel(), 2 * nsamp), dtype=torch.uint8, device="cuda", generator=gen)
    stream.synchronize()
    first_host = (np.ascontiguousarray(base_host[: min(ns, 64), : 2 * nsamp]) if distinct >= min(ns, 64) and not mixed_percent
                  else batches[0][: min(ns, 64)].cpu().numpy())
    return batches, np.ascontiguousarray(first_host), t_gen


def read_ceiling(pkg, batches, local_rank, passes=60):
    """GB/s of a read-only LDS-DMA stream with the headline kernel's access pattern over the same rotated batches (a hook of the library:
    include/sdrfm_dev.h sdrfm_debug_read_ceiling) — the measured ceiling beside the 8 TB/s specification.  ~2 ms of GPU time."""
    import ctypes as C
    lib = pkg.load_library()
    ptrs = (C.c_void_p * len(batches))(*[b.data_ptr() for b in batches])
    out = C.c_double()
    rc = lib.sdrfm_debug_read_ceiling(local_rank, ptrs, len(batches), batches[0].numel(), passes, C.byref(out))
    return out.value if rc == 0 else None


def pick_batches(args, bytes_per_batch):
    if args.batches > 0:
        return args.batches
    return max(3, -(-int(1.5 * L3_BYTES) // int(bytes_per_batch)) + 1)   # (nb - 1) batches between two uses of one > 1.5 x L3


PCM_CALL = {"on": False}                                         # --pcm-call pcm | both: every call through sdrfm_process_batch_pcm
DIST_DEVICE = {"d": "cuda"}         # where the ranks' bookkeeping tensors live (the CPU test of that bookkeeping, under gloo, sets "cpu")
TIMED_WITH_BARRIER = {"s": None}   # (N > 1: the last timed region's wall clock with the closing barrier inside the bracket, MAX over ranks)


def timed(torch, dist, use_dist, stream, step, steps, finish=None, rest=0.0):
    """K back-to-back steps between ONE pair of HIP events on the launch stream; returns (max-over-ranks wall seconds, event span / K in ms).
    N = 1: the wall clock runs from behind the opening synchronize to behind the closing one.  N > 1: barrier + synchronize open the bracket on
    every rank; a rank's clock stops at its OWN closing synchronize, the closing barrier follows, and the figure is the MAX over ranks (the skew
    between ranks leaving the opening barrier is therefore not in it; `ms_per_step_with_closing_barrier` in the line is the common-end reading).
    `finish` (overlapped calls: sdrfm_flush) is called after the
    last step, before the closing event: it puts the launch stream behind every call.  `rest` seconds of idle GPU first (the secondary
    regions: each starts from the same rested state instead of inheriting the previous region's power / clock state — a serial region run
    right behind 100 overlapped calls measured 37 us per call against 27 - 29 us a few milliseconds later)."""
    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    fence()
    if rest > 0.0:
        time.sleep(rest)
        fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(steps):
        step(i)
    if finish is not None:
        finish()
    ev1.record(stream)
    t_poll = time.perf_counter()
    while not ev1.query() and time.perf_counter() - t_poll < 30.0:   # (the closing synchronize then returns at once: a blocking wait's wake-up latency is not part of K steps)
        pass
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        # Every rank's clock stops at its OWN closing synchronize; the closing barrier follows, and the figure is the MAX over ranks — the time until the slowest
        # rank was done, which is what a barrier inside the bracket measures plus the barrier's own latency (through RCCL: ~0.6 ms, more than the 20 steps of the
        # driver's run take: it would halve `value` at every N > 1 and say nothing about the K steps).  TIMED_WITH_BARRIER keeps the other figure for the line.
        dist.barrier()
        torch.cuda.synchronize()
        TIMED_WITH_BARRIER["s"] = time.perf_counter() - t0
        t = torch.tensor([elapsed, TIMED_WITH_BARRIER["s"]], dtype=torch.float64, device=DIST_DEVICE["d"])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, TIMED_WITH_BARRIER["s"] = float(t[0].item()), float(t[1].item())
    return elapsed, ev0.elapsed_time(ev1) / steps


def gather_per_rank(torch, dist, world, x, device="cuda"):
    """every rank's figure on every rank (N > 1: `roofline.per_rank_kernel_ms`)"""
    t = torch.tensor([x], dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    return [round(float(v.item()), 4) for v in gathered]


def scaling_fields(world, rccl_world, ns, nsamp, steps, elapsed, elapsed_with_barrier, use_dist):
    """the fields of the JSON line that depend on the number of ranks: whole-job throughput over ALL ranks' streams / the max-over-ranks time"""
    out = {"value": round(float(world) * ns * nsamp * steps / elapsed / 1e6, 1), "unit": "MSamples/s", "n_gpus": world, "rccl_world": rccl_world,
           "ms_per_step": round(elapsed / steps * 1e3, 4), "scaling": "weak"}
    if use_dist and elapsed_with_barrier:
        out["timing"] = ("N ranks through RCCL: the K steps from the opening barrier + synchronize to each rank's own closing synchronize, MAX over ranks; the closing barrier "
                         "follows the clock — its own latency is in ms_per_step_with_closing_barrier, not in value")
        out["ms_per_step_with_closing_barrier"] = round(elapsed_with_barrier / steps * 1e3, 4)
    return out


def read_basis(samples_per_launch, ms):
    """fraction of the 8 TB/s HBM-READ roofline: samples x 2 B / time — the basis BASELINE.md section 2 defines the >= 70 % target on (the audio writes are not counted)"""
    return round(samples_per_launch * 2.0 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms else None


def steady_regions(torch, stream, step, finish=None, regions=10, steps=300):
    """What a consumer that runs for seconds sees: `regions` regions of `steps` calls back to back (no rest between them), each between one pair of
    HIP events on the launch stream -> ms per call of every region.  From a rested GPU the first regions run slower than the later ones: the
    power management dips between ~1 ms and ~30 ms after a load starts (profiles/r05_clock_transient.txt) and then settles; a 20-step region sits
    before the dip, a 300-step region from rest inside it, the last regions of this series behind it."""
    out = []
    for _ in range(regions):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(stream)
        for i in range(steps):
            step(i)
        if finish is not None:
            finish()
        ev1.record(stream)
        ev1.synchronize()
        out.append(ev0.elapsed_time(ev1) / steps)
    return out


def per_launch_events(torch, stream, step, n):
    """Untimed pass: one HIP event pair around each of n launches on the launch stream -> list of ms."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i, (a, b) in enumerate(evs):
        a.record(stream)
        step(i)
        b.record(stream)
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                                         # never returns
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # under torch.distributed.run: always go through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    rccl_world = dist.get_world_size() if dist.is_initialized() else 1

    pkg = importlib.import_module("stm32f7-rtlsdr_amd")
    if args.workload == "wbfm":
        return main_wbfm(args, pkg, world, rank, local_rank, use_dist, rccl_world)
    if args.workload == "spectrum":
        return main_spectrum(args, pkg, world, rank, local_rank, use_dist, rccl_world)
    D, Da = args.fir_decim, (8 if args.fir_decim == 8 else 5)
    fs = {8: 2.048e6, 10: 2.4e6, 16: 3.2e6}[D]
    ns = args.streams_per_gpu if args.streams_per_gpu > 0 else (256 if world == 1 else 512)
    cfg_name = "configs[2]" if ns == 256 else ("configs[3] share (4096 streams / 8 GPUs)" if ns == 512 else "configs[2]-shaped")
    nsamp = int(round(args.seconds * fs))
    nbytes = 2 * nsamp
    h, g = pkg.default_config(args.fir_taps, fs=fs, fir_decim=D, audio_taps=32, audio_decim=Da)

    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, device=local_rank,
                                  dev_library=args.dev_library, bit_exact=args.bit_exact))
    stream = torch.cuda.Stream()
    dm.set_stream(stream.cuda_stream)
    nb = pick_batches(args, ns * nbytes)
    batches, iq_host, t_gen = make_batches(torch, pkg, stream, ns, nsamp, fs, rank, args.distinct, nb, mode=args.iq_class)
    n_audio_max = nsamp // D // Da + 1
    with torch.cuda.stream(stream):
        audio = torch.zeros((ns, n_audio_max), dtype=torch.float32, device="cuda")
    stream.synchronize()

    e2e = None
    if args.end_to_end:
        # rank 0 owns the whole batch; each step = C1 scatter -> local hot path -> C2 gather, all on `stream`
        if not dist.is_initialized():
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        with torch.cuda.stream(stream):
            iq_all = batches[0].repeat(world, 1) if rank == 0 else None
        stream.synchronize()
        e2e = {"iq_all": iq_all, "total": world * ns}

    last = {"n": 0}

    pcm_sink = pcm_bufs = None
    PCM_CALL["on"] = args.pcm_call != "none"
    if args.pcm_call != "none":                                   # (profiling aid: every call leaves the PCM — three buffers in turn — instead of / beside the audio)
        alpha_p, gain_p = float(pkg.load_library().sdrfm_pcm_alpha(48000.0, 75e-6)), float(32767.0 / (2 * np.pi * 75e3 / (fs / D)))
        pcm_sink = pkg.PcmSink(ns, alpha_p, gain_p, device=local_rank)
        with torch.cuda.stream(stream):
            pcm_bufs = [torch.zeros((ns, 2 * n_audio_max + 2), dtype=torch.int16, device="cuda") for _ in range(3)]
            audio3 = [audio, torch.zeros_like(audio), torch.zeros_like(audio)]
        stream.synchronize()
    pcm_calls = {"n": 0}

    def call_pcm(batch, ovl):
        k = pcm_calls["n"] % 3
        pcm_calls["n"] += 1
        return dm.process_batch_pcm_device(pcm_sink, batch, audio3[k] if args.pcm_call == "both" else None, pcm_bufs[k], overlap=ovl)

    def step_rot(i):
        if pcm_sink is not None:
            last["n"] = call_pcm(batches[i % nb], False)
            return
        if e2e is None:
            last["n"] = dm.process_batch_device(batches[i % nb], audio)
            return
        with torch.cuda.stream(stream):
            local = pkg.fanout.scatter_streams(e2e["iq_all"], e2e["total"], nbytes, batches[0].device)
            last["n"] = dm.process_batch_device(local, audio)
            pkg.fanout.gather_audio(audio[:, :last["n"]], e2e["total"])

    def step_res(i):
        last["n"] = dm.process_batch_device(batches[0], audio)

    # SDRFM_F_OVERLAP (include/sdrfm.h): consecutive calls may run concurrently on the device — a call's start-up under the previous
    # call's tail.  What the flag asks of the caller holds here by construction: the previous call's input batch stays intact (the loop
    # rotates over nb >= 3 resident batches) and the audio goes to two buffers in turn.
    overlap = e2e is None and not args.no_overlap and nb >= 3
    if overlap:
        with torch.cuda.stream(stream):
            audio_pair = [audio, torch.zeros_like(audio)]
        stream.synchronize()

    ovl_calls = {"n": 0}

    def step_ovl(i):
        # the two audio buffers are taken in turn ACROSS regions, as a running caller's are: a region that started again with the buffer the
        # previous region's last call wrote (i & 1 restarting at 0) had its first call made serially by the library (two consecutive calls
        # must not write the same audio buffer), and the next few waited on the handle's stream for it — 2 us per call of a 20-call region
        # (profiles/r05_driver_command_trace_before.json)
        if pcm_sink is not None:
            last["n"] = call_pcm(batches[i % nb], True)
            return
        last["n"] = dm.process_batch_device(batches[i % nb], audio_pair[ovl_calls["n"] & 1], overlap=True)
        ovl_calls["n"] += 1

    if args.iq_class != "fm":
        # a capture that has been running for a while: the library's per-stream statistics (windows of 16 calls, read back asynchronously) have settled
        for i in range(96):
            (step_ovl if overlap else step_rot)(i)
            if i % 8 == 7:
                time.sleep(0.002)
        dm.flush()
        dm.synchronize()
    for i in range(args.warmup):
        (step_ovl if overlap else step_rot)(i)
    dm.flush()
    # Timed region: K back-to-back launches between ONE pair of HIP events on the launch stream (an event pair around every
    # launch costs ~6 us of idle GPU per step).  kernel_ms_avg = event span / K is therefore the average launch duration
    # INCLUDING any inter-launch gap; per-launch event timings come from a short untimed pass afterwards.
    kernel_ms_ovl = kernel_ms_ovl_sus = None
    served_by = None
    routed_timed = routed_kernel = None
    serial_regions = None
    if overlap:
        elapsed, kernel_ms_ovl = timed(torch, dist, use_dist, stream, step_ovl, args.steps, finish=dm.flush)
        elapsed_with_barrier = TIMED_WITH_BARRIER["s"]
        served_by = dm.kernel_name
        routed_timed, routed_kernel = dm.route(), served_by      # (the assignment the timed region ran with: a stream is tried on design Q again every 1024 calls)
        if "overlapped" not in served_by:                        # a handle the matrix-pipe kernel does not serve (--bit-exact, other taps): the
            overlap = False                                      # library makes the calls one after the other; time them as such (below)
    if overlap:
        # the roofline figure: the same K calls made one after the other (what rocprofv3 --kernel-trace reports as the kernel's duration),
        # from a rested GPU and behind the same warm-up as the timed region
        # — taken five times when K is short: the clock state a short region meets varies from region to region on one box (26.4 - 31.3 us per
        # call over 20 steps in seven consecutive regions of one process, whatever the rest between them; the overlapped regions do not show
        # this); `frac` is the MEDIAN region, all five are listed and the best one is given beside it
        serial_regions = []
        for _ in range(5 if args.steps < 100 else 1):
            time.sleep(REST_S)
            for i in range(args.warmup):
                step_rot(i)
            serial_regions.append(timed(torch, dist, use_dist, stream, step_rot, args.steps)[1])
        kernel_ms_avg = sorted(serial_regions)[len(serial_regions) // 2]
    else:
        elapsed, kernel_ms_avg = timed(torch, dist, use_dist, stream, step_rot, args.steps)
        elapsed_with_barrier = TIMED_WITH_BARRIER["s"]
        routed_timed, routed_kernel = dm.route(), dm.kernel_name
    n_audio = last["n"]
    kernel_ms = per_launch_events(torch, stream, step_rot, min(args.steps, 20))
    # sustained figure: at least 300 back-to-back steps (a short timed region runs at the boost clock)
    if args.steps >= 300 or e2e is not None:
        kernel_ms_sus, sus_steps = kernel_ms_avg, args.steps
    else:
        sus_steps = 300
        _, kernel_ms_sus = timed(torch, dist, use_dist, stream, step_rot, sus_steps, rest=REST_S)
    if overlap:
        _, kernel_ms_ovl_sus = (None, kernel_ms_ovl) if args.steps >= 300 else timed(torch, dist, use_dist, stream, step_ovl, 300, finish=dm.flush, rest=REST_S)
    # steady state: ten regions of 300 calls back to back; the median of the last five (single-GPU runs: a scaling run's ranks are not held together here)
    steady_ser = steady_ovl = None
    if e2e is None and not use_dist and not args.no_steady:
        time.sleep(REST_S)
        steady_ser = steady_regions(torch, stream, step_rot)
        if overlap:
            time.sleep(REST_S)
            steady_ovl = steady_regions(torch, stream, step_ovl, finish=dm.flush)
    # what a consumer of the audio gets: the same overlapped calls made through sdrfm_process_batch_pcm — the board's sink format
    # (de-emphasis, int16 L = R) out of the demodulator's own launch —, steady state, with and without the float audio stored beside the PCM
    consumer = None
    if steady_ovl and overlap and not args.dev_library and not args.no_consumer_leg and pcm_sink is None:
        alpha, gain = float(pkg.load_library().sdrfm_pcm_alpha(48000.0, 75e-6)), float(32767.0 / (2 * np.pi * 75e3 / (fs / D)))
        with torch.cuda.stream(stream):
            pcm_ring = [torch.zeros((ns, 2 * n_audio_max + 2), dtype=torch.int16, device="cuda") for _ in range(3)]
            audio_ring = audio_pair + [torch.zeros_like(audio)]
        stream.synchronize()
        with pkg.PcmSink(ns, alpha, gain, device=local_rank) as sink:
            cc = {"n": 0, "names": set()}

            def step_pcm(with_audio):
                def f(i):
                    k = cc["n"] % 3
                    dm.process_batch_pcm_device(sink, batches[i % nb], audio_ring[k] if with_audio else None, pcm_ring[k], overlap=True)
                    cc["n"] += 1
                return f
            time.sleep(REST_S)
            both = steady_regions(torch, stream, step_pcm(True), finish=dm.flush)
            name_pcm = dm.kernel_name
            only = steady_regions(torch, stream, step_pcm(False), finish=dm.flush)
            dm.synchronize()
            sink_ok = sink.synchronize_status() == 0
        consumer = {"call": "sdrfm_process_batch_pcm (SDRFM_F_OVERLAP), three PCM buffers in turn; steady = median of the last five of ten regions of 300 calls",
                    "kernel": name_pcm, "ms_per_call_pcm_and_audio": round(float(np.median(both[5:])), 5), "ms_per_call_pcm_only": round(float(np.median(only[5:])), 5),
                    "ms_per_call_demodulator_alone": round(float(np.median(steady_ovl[5:])), 5),
                    "ms_per_call_regions_300": {"pcm_and_audio": [round(x, 4) for x in both], "pcm_only": [round(x, 4) for x in only]},
                    "sink_chain_ok": sink_ok,
                    "note": "the sink's chain runs inside the demodulator's launch (csrc/sdrfm_sink_chain.h); PCM within 1 LSB of the host routine's (tests/test_pcm_sink_gpu.py); "
                            "the stand-alone sink kernel behind every call: profiles/r06_sink.txt"}
    # the north-star's literal design beside the default one: a SDRFM_CFG_BIT_EXACT handle — fp32 fmaf chains on the vector pipe, no matrix
    # instruction, bit-identical to the definition's chains — over the same rotated batches: K serial calls from rest (median of five regions when K is short) and
    # the steady series.  A second handle; nothing of the default handle's figures depends on it.
    bit_exact_leg = None
    if e2e is None and not use_dist and not args.bit_exact and not args.dev_library and not args.no_bit_exact_leg and args.iq_class == "fm":
        time.sleep(REST_S)
        dx = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, device=local_rank, bit_exact=True))
        dx.set_stream(stream.cuda_stream)

        def step_bx(i):
            dx.process_batch_device(batches[i % nb], audio)
        regs = []
        for _ in range(5 if args.steps < 100 else 1):
            time.sleep(REST_S)
            for i in range(args.warmup):
                step_bx(i)
            regs.append(timed(torch, dist, False, stream, step_bx, args.steps)[1])
        bx_ms = sorted(regs)[len(regs) // 2]
        bx_steady = None
        if not args.no_steady:
            time.sleep(REST_S)
            bx_steady = steady_regions(torch, stream, step_bx)
        bit_exact_leg = {"kernel": dx.kernel_name, "ms": bx_ms, "regions": regs, "steady": bx_steady}
        dx.synchronize()
        dx.set_stream(None)
        dx.close()
    # second, labelled figure: ONE resident input batch (fits the 256 MiB Infinity Cache) — what round 1 reported as `value`
    res_steps = min(args.steps, 100)
    for i in range(min(args.warmup, 10)):
        step_res(i)
    elapsed_res, kernel_ms_res = timed(torch, dist, use_dist, stream, step_res, res_steps) if e2e is None else (None, None)

    # measured read ceiling (a read-only stream with the kernel's own access pattern over the same rotated batches), and how much of the
    # timed work went through the matrix-pipe kernel's repair path
    time.sleep(REST_S)
    peak_measured = read_ceiling(pkg, batches, local_rank) if e2e is None else None
    guard = dm.q_guard()
    per_rank = gather_per_rank(torch, dist, world, kernel_ms_avg) if use_dist else None

    ok = None
    if args.check and rank == 0:
        from oracle.oracle import Oracle
        dm.reset()
        step_res(0)
        dm.synchronize()
        got = audio[:, :n_audio].cpu().numpy()
        ok = True
        for s in (0, min(ns, iq_host.shape[0]) // 2, min(ns, iq_host.shape[0]) - 1):
            want = Oracle(h, g).process(iq_host[s])
            err = np.max(np.abs(got[s] - want) / np.maximum(np.abs(want), 1.0))
            ok = ok and bool(err <= 1e-5)

    if rank == 0:
        sf = scaling_fields(world, rccl_world, ns, nsamp, args.steps, elapsed, elapsed_with_barrier, use_dist)
        value = sf["value"]
        samples_per_launch = ns * nsamp
        alg_bytes = samples_per_launch * 2.0 + ns * n_audio * 4.0        # 2 B in + 4/(D*Da) B out per IQ sample
        if args.pcm_call == "both":
            alg_bytes += ns * n_audio * 4.0                             # (--pcm-call both: the int16 pair beside the float; 'pcm': instead of it — the same 4 B)
        achieved = alg_bytes / (kernel_ms_avg * 1e-3) / 1e9
        traffic = latest_traffic(dm.kernel_name, alg_bytes)
        res = {
            "metric": "IQ MSamples/s through FIR+FM-demod+resample",
            **{k: sf[k] for k in ("value", "unit", "n_gpus", "rccl_world")}, "steps": args.steps, "warmup": args.warmup,
            **{k: sf[k] for k in ("ms_per_step", "timing", "ms_per_step_with_closing_barrier") if k in sf},
            "higher_is_better": True, "scaling": sf["scaling"], "vs_baseline": None,
            "dtype": ("f32 audio; K2 = u8 x 24-bit fixed-point taps on the i8 matrix pipe, exact i32 sums, one f32 recombination (design Q); K3 / K4 f32; "
                      "ill-conditioned phases recomputed with the f32 fmaf chain" if dm.kernel_name.startswith("fast-q") else "f32 (fmaf chains throughout)"),
            "data": ("synthetic" if args.iq_class == "fm" else
                     "synthetic, uniform random bytes (SURVEY.md 8d's worst-case class: NOT the headline input)" if args.iq_class == "random" else
                     "synthetic, %s per cent of the streams uniform random bytes, the others the FM test signal (NOT the headline input)" % args.iq_class.split(":")[1]),
            "mode": "end-to-end (RCCL scatter of IQ from rank 0 + gather of audio every step)" if args.end_to_end else
                    ("compute-only (IQ resident per GPU); timed calls made with SDRFM_F_OVERLAP: consecutive calls may run concurrently on the device"
                     if overlap else "compute-only (IQ resident per GPU); calls one after the other"),
            "config": {"workload": ("BASELINE %s: " if D == 10 else "(NOT a BASELINE config: the %s shape at another front-end rate) ") % cfg_name +
                                   "%d concurrent %.3f MS/s uint8 IQ streams per GPU x %.1f s (%d B each), "
                                   "%d-tap FIR /%d + FM discriminator + %d-tap /%d; device-resident, streams sharded "
                                   "across GPUs with no collective; timed loop rotates over %d input batches = %.0f MB per GPU (> 256 MiB "
                                   "L3: cold HBM reads)" % (ns, fs / 1e6, args.seconds, nbytes, args.fir_taps, D, len(g), Da, nb, nb * ns * nbytes / 1e6),
                       "streams_per_gpu": ns, "bytes_per_stream": nbytes, "fir_taps": args.fir_taps, "kernel": dm.kernel_name,
                       "input_batches_rotated": nb, "input_bytes_rotated_per_gpu": nb * ns * nbytes,
                       "bytes_per_sample_algorithmic": round(alg_bytes / samples_per_launch, 4),
                       "GB_per_s_input": round(value * 2e6 / 1e9, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "frac_sustained": round(alg_bytes / (kernel_ms_sus * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                         "sustained_steps": sus_steps, "kernel_ms_sustained": round(kernel_ms_sus, 4),
                         "peak_measured": round(peak_measured, 1) if peak_measured else None,
                         "frac_of_measured": round(achieved / peak_measured, 4) if peak_measured else None,
                         "frac_sustained_of_measured": round(alg_bytes / (kernel_ms_sus * 1e-3) / 1e9 / peak_measured, 4) if peak_measured else None,
                         "peak_measured_note": "read-only LDS-DMA stream with this kernel's access pattern (12 one-wave workgroups per CU, 5 KiB in flight each, nt) over the same rotated batches, 60 passes in this run (sdrfm_debug_read_ceiling)",
                         "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                         "traffic_source": ("profiles/%s (rocprofv3 --pmc TCC_EA0 read/write request passes at commit %s; same kernel, same workload size)" % (traffic["file"], traffic.get("commit", "?"))) if traffic else None,
                         "kernel": dm.kernel_name, "kernel_ms_avg": round(kernel_ms_avg, 4),
                         "kernel_ms_isolated_avg": round(float(np.mean(kernel_ms)), 4),
                         "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         **({"frac_steady": round(alg_bytes / (float(np.median(steady_ser[5:])) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                             "kernel_ms_regions_300": [round(x, 4) for x in steady_ser],
                             "steady_note": "ten regions of 300 calls back to back from a rested GPU; frac_steady = the median of the last five (the state a consumer "
                                            "that runs for seconds sees); frac_sustained = ONE 300-call region from rest, which sits inside the power management's dip "
                                            "between ~1 and ~30 ms after a load starts (profiles/r05_clock_transient.txt)"} if steady_ser else {}),
                         **({"kernel_ms_avg_regions": [round(x, 4) for x in serial_regions],
                             "frac_best_region": round(alg_bytes / (min(serial_regions) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                            if serial_regions and len(serial_regions) > 1 else {})},
            "gen_seconds": round(t_gen, 2),
        }
        # The >= 70 % target of BASELINE.md section 2 is defined on the HBM-READ basis (2 B per IQ sample; the audio writes are not counted): every
        # fraction above again on that basis, and whether the target is met — judged on the steady overlapped calls (what a consumer that runs for
        # seconds gets through the API's fastest way), or on the timed region's overlapped calls when the steady series was not taken.
        rl = res["roofline"]
        rb = {"bytes_per_launch": samples_per_launch * 2.0, "frac": read_basis(samples_per_launch, kernel_ms_avg),
              "frac_sustained": read_basis(samples_per_launch, kernel_ms_sus),
              "frac_steady": read_basis(samples_per_launch, float(np.median(steady_ser[5:]))) if steady_ser else None,
              "note": "samples x 2 B / time / 8 TB/s — the basis the >= 70 % target is defined on; `frac` etc. above count the audio writes too (2.08 B per sample)"}
        if overlap:
            rb["overlapped_calls"] = {"frac": read_basis(samples_per_launch, kernel_ms_ovl), "frac_sustained": read_basis(samples_per_launch, kernel_ms_ovl_sus),
                                      "frac_steady": read_basis(samples_per_launch, float(np.median(steady_ovl[5:]))) if steady_ovl else None}
        rl["read_basis"] = rb
        rl["frac_read_basis"] = rb["frac"]
        judged = (rb.get("overlapped_calls", {}).get("frac_steady") or rb.get("overlapped_calls", {}).get("frac") or rb["frac_steady"] or rb["frac"])
        rl["target_70pct_read_roofline_met"] = bool(judged is not None and judged >= 0.70) if (D == 10 and args.iq_class == "fm" and not args.bit_exact) else None
        rl["target_judged_on"] = {"frac_read_basis": judged, "which": ("overlapped calls, steady state" if rb.get("overlapped_calls", {}).get("frac_steady") else
                                                                      "overlapped calls, the timed region" if rb.get("overlapped_calls", {}).get("frac") else
                                                                      "serial calls, steady state" if rb["frac_steady"] else "serial calls, the timed region")}
        if consumer:
            res["consumer_loop"] = consumer
        if bit_exact_leg:
            bx = bit_exact_leg
            res["bit_exact_kernel"] = {
                "handle": "SDRFM_CFG_BIT_EXACT: the north-star's literal design (fp32 fmaf chains on the vector pipe, no matrix instruction), bit-identical to the definition's chains; NOT the default path",
                "kernel": bx["kernel"], "kernel_ms_avg": round(bx["ms"], 4), "kernel_ms_avg_regions": [round(x, 4) for x in bx["regions"]],
                "value": round(samples_per_launch / (bx["ms"] * 1e-3) / 1e6, 1), "unit": "MSamples/s (serial calls, event span / K)",
                "frac": round(alg_bytes / (bx["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "frac_read_basis": read_basis(samples_per_launch, bx["ms"]),
                **({"frac_steady": round(alg_bytes / (float(np.median(bx["steady"][5:])) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "frac_steady_read_basis": read_basis(samples_per_launch, float(np.median(bx["steady"][5:]))),
                    "kernel_ms_regions_300": [round(x, 4) for x in bx["steady"]]} if bx["steady"] else {})}
        routed = routed_timed
        if routed is not None and (args.iq_class != "fm" or int(routed.sum())):
            res["routing"] = {"streams_on_bit_exact_kernels": int(routed.sum()), "streams": ns, "kernels_timed_region": routed_kernel,
                              "note": "per-stream routing (DESIGN.md 4.Q), as the timed region ended: streams whose windows of design-Q calls were mostly repair "
                                      "work are served by the bit-exact kernels — design B workgroups inside design Q's launch (k_mix) where both designs have an instance, "
                                      "a launch of their own ahead of design Q's for the others"}
        if guard:
            res["guard"] = {"guard_r": round(guard["guard_r"], 4), "pi_minus_guard_a": round(3.141592653589793 - guard["guard_a"], 7),
                            "lanes_repaired_since_create": guard["lanes"], "repair_passes_since_create": guard["passes"],
                            "note": "design Q's conditioning guard (DESIGN.md 4.Q): lanes (pairs of discriminator outputs) recomputed with the definition's fmaf chain, over every call this handle served in this run"}
        if per_rank:
            res["roofline"]["per_rank_kernel_ms"] = {"min": min(per_rank), "max": max(per_rank), "all": per_rank}
        if overlap:
            # `roofline` above: calls one after the other = the kernel's own duration.  This: the timed region itself (overlapped calls);
            # event span / K is then shorter than a kernel's duration, because consecutive kernels run side by side.
            res["roofline"]["overlapped_calls"] = {
                "kernel": served_by, "ms_per_call": round(kernel_ms_ovl, 4),
                "frac": round(alg_bytes / (kernel_ms_ovl * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "frac_sustained": round(alg_bytes / (kernel_ms_ovl_sus * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                **({"frac_steady": round(alg_bytes / (float(np.median(steady_ovl[5:])) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "ms_per_call_regions_300": [round(x, 4) for x in steady_ovl]} if steady_ovl else {}),
                # against the read ceiling measured in this run (roofline.peak_measured: what a read-only stream with the kernel's access pattern reaches on this box)
                **({"frac_of_measured": round(alg_bytes / (kernel_ms_ovl * 1e-3) / 1e9 / peak_measured, 4)} if peak_measured else {}),
                **({"frac_steady_of_measured": round(alg_bytes / (float(np.median(steady_ovl[5:])) * 1e-3) / 1e9 / peak_measured, 4)} if (peak_measured and steady_ovl) else {}),
                "note": "bytes per call / (event span of the timed region / K); `value` is measured on these calls"}
        if args.bit_exact:
            res["handle"] = "SDRFM_CFG_BIT_EXACT (fmaf-chain kernels only: NOT the default path)"
        if args.dev_library:
            res["library"] = "libsdrfm_dev.so (development build: NOT the product figure)"
        if elapsed_res is not None:
            res["resident_input"] = {"value": round(float(world) * ns * nsamp * res_steps / elapsed_res / 1e6, 1), "unit": "MSamples/s",
                                     "kernel_ms_avg": round(kernel_ms_res, 4), "steps": res_steps,
                                     "frac_of_hbm_peak": round(alg_bytes / (kernel_ms_res * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                     "note": "same kernel re-reading ONE %.0f MB batch, which the 256 MiB Infinity Cache can hold; not the headline" % (ns * nbytes / 1e6)}
        if ok is not None:
            res["parity_ok"] = ok
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(pkg, h, g, iq_host, args.cpu_seconds, os.cpu_count() or 1)
        print(json.dumps(res), flush=True)
    dm.set_stream(None)
    dm.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_wbfm(args, pkg, world, rank, local_rank, use_dist, rccl_world):
    """BASELINE configs[4]: 3.2 MS/s IQ, 128-tap prototype, 16 bands, per-band FM demod, 6/25 resampler; 128 streams/GPU."""
    import torch
    import torch.distributed as dist
    fs, ns = 3.2e6, (args.streams_per_gpu if args.streams_per_gpu > 0 else 128)
    nsamp = int(round(args.seconds * fs))
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, device=local_rank, max_bytes_per_call=2 * nsamp))
    kname = dm.kernel_name
    stream = torch.cuda.Stream()
    dm.set_stream(stream.cuda_stream)
    nb = pick_batches(args, ns * 2 * nsamp)
    batches, _, _ = make_batches(torch, pkg, stream, ns, nsamp, fs, rank, args.distinct, nb)
    cap = dm.audio_count(2 * nsamp) + 8
    with torch.cuda.stream(stream):
        audio = torch.zeros((ns, 16, cap), dtype=torch.float32, device="cuda")
    stream.synchronize()
    last = {"n": 0}

    def step(i):
        last["n"] = dm.process_batch_device(batches[i % nb], audio)

    for i in range(args.warmup):
        step(i)
    elapsed, ms = timed(torch, dist, use_dist, stream, step, args.steps)   # ms = launch duration incl. inter-launch gap (see main())
    n_audio = last["n"]
    if rank == 0:
        alg = ns * nsamp * 2.0 + ns * 16 * n_audio * 4.0
        tr = latest_traffic(kname, alg)
        res = {"metric": "IQ MSamples/s through FIR+FM-demod+resample", "value": round(world * ns * nsamp * args.steps / elapsed / 1e6, 1),
               "unit": "MSamples/s", "n_gpus": world, "rccl_world": rccl_world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32 (fmaf chains and the fixed radix-2 butterfly graph of the spec, bit-identical to the oracle up to the discriminator)", "data": "synthetic",
               "config": {"workload": "BASELINE configs[4]: %d x 3.2 MS/s uint8 IQ streams per GPU x %.1f s, 128-tap prototype, 16-band polyphase "
                                      "channelizer + per-band FM demod + 6/25 resampler -> 16 x 48 kHz; %d input batches rotated (%.0f MB, cold HBM reads)"
                                      % (ns, args.seconds, nb, nb * ns * 2 * nsamp / 1e6),
                          "streams_per_gpu": ns, "bytes_per_stream": 2 * nsamp, "kernel": kname, "input_batches_rotated": nb},
               "roofline": bound_block("valu", kname, ms, alg, (tr or {}).get("hbm_bytes_per_launch"))}
        print(json.dumps(res), flush=True)
    dm.set_stream(None)
    dm.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main_spectrum(args, pkg, world, rank, local_rank, use_dist, rccl_world):
    """FFT view of the BASELINE configs[2] buffers: 256 x 0.1 s of 2.4 MS/s IQ per GPU -> averaged power spectra."""
    import numpy as np
    import torch
    import torch.distributed as dist
    fs, ns, nfft = 2.4e6, (args.streams_per_gpu if args.streams_per_gpu > 0 else 256), args.nfft
    nsamp = int(round(args.seconds * fs))
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, device=local_rank, max_bytes_per_call=2 * nsamp, dev_library=args.dev_library))
    stream = torch.cuda.Stream()
    sv.set_stream(stream.cuda_stream)
    nb = pick_batches(args, ns * 2 * nsamp)
    batches, _, _ = make_batches(torch, pkg, stream, ns, nsamp, fs, rank, args.distinct, nb)
    with torch.cuda.stream(stream):
        power = torch.zeros((ns, nfft), dtype=torch.float32, device="cuda")
    stream.synchronize()
    last = {"n": 0}

    def step(i):
        last["n"] = sv.process_batch_device(batches[i % nb], power)

    for i in range(args.warmup):
        step(i)
    elapsed, ms = timed(torch, dist, use_dist, stream, step, args.steps)
    frames = last["n"]
    if rank == 0:
        alg = ns * frames * nfft * 2.0 + ns * nfft * 4.0
        kname = sv.kernel_name
        res = {"metric": "IQ MSamples/s through the windowed-FFT spectrum view", "value": round(world * ns * frames * nfft * args.steps / elapsed / 1e6, 1),
               "unit": "MSamples/s", "n_gpus": world, "rccl_world": rccl_world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32 (the spec's fixed radix-2 butterfly graph: bit-identical to the oracle)", "data": "synthetic",
               "config": {"workload": "spectrum view (SURVEY 8f-3) of BASELINE configs[2] buffers: %d x 2.4 MS/s uint8 IQ streams per GPU x %.1f s, "
                                      "%d-point Hann FFT, %d frames averaged per stream; %d input batches rotated (cold HBM reads)" % (ns, args.seconds, nfft, frames, nb),
                          "streams_per_gpu": ns, "bytes_per_stream": 2 * nsamp, "kernel": kname, "input_batches_rotated": nb},
               "roofline": bound_block("valu" if "chain" in kname else "lds", kname, ms, alg, (latest_traffic(kname, alg) or {}).get("hbm_bytes_per_launch"))}
        print(json.dumps(res), flush=True)
    sv.set_stream(None)
    sv.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
