#!/usr/bin/env python3
"""bench.py — IQ MSamples/s through FIR + FM-demod + resample on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (sdrfm_process_batch on device-resident buffers: one fused kernel launch) over one
batch of synthetic IQ: by default BASELINE configs[2] — 256 concurrent 2.4 MS/s streams x 0.1 s (480 000 B each),
64-tap FIR /10, FM discriminator, 32-tap /5 audio resampler.  With --gpus N (launched by torch.distributed.run) every
rank owns its own 256 streams on its own GPU (streams are independent: no data-path collective, weak scaling).

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).  The GPU legs never touch oracle/; only the
`cpu_baseline` leg (rank 0, N=1) times the scalar-C oracle on a bounded sample of the same workload.
"""
import argparse
import glob
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--streams-per-gpu", type=int, default=256)
    ap.add_argument("--seconds", type=float, default=0.1, help="capture length per stream per step")
    ap.add_argument("--fir-taps", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=0, help="distinct synthetic streams (0 = all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--check", action="store_true", help="also verify a few streams against the oracle (not timed)")
    ap.add_argument("--workload", choices=["fm", "wbfm", "spectrum"], default="fm",
                    help="fm = BASELINE configs[2] (default, the headline); wbfm = configs[4] channelizer path, 128 streams/GPU; "
                         "spectrum = FFT view (SURVEY 8f-3) of the configs[2] buffers")
    ap.add_argument("--nfft", type=int, default=1024, help="spectrum workload: FFT length")
    ap.add_argument("--end-to-end", action="store_true",
                    help="fm workload: every step also scatters the IQ batch from rank 0 to all ranks and gathers the audio back "
                         "over RCCL (SURVEY 8e C1/C2); reported separately from the compute-only default")
    return ap.parse_args()


def latest_traffic(kernel_name):
    """HBM bytes per launch from the newest committed PMC summary (profiles/traffic_r*.json), if it is for this kernel."""
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json"))):
        try:
            with open(fn) as f:
                t = json.load(f)
            if t.get("kernel_name") == kernel_name:
                best = t
        except Exception:
            pass
    return best


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
    out = {"value": round(v1, 3), "unit": "MSamples/s", "cores": 1, "kind": "port",
           "sample": "%d stream-chunks of the workload (%.1f s x 2.4 MS/s each), scalar-C oracle, 1 thread, %.0f s" % (n1, nsamp / 2.4e6, seconds)}
    if threads > 1:
        vn, nn = run(threads, max(3.0, seconds / 2))
        out["all_cores"] = {"value": round(vn, 3), "cores": threads, "stream_chunks": nn}
    return out


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ     # under torch.distributed.run: always go through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("stm32f7-rtlsdr_amd")
    if args.workload == "wbfm":
        return main_wbfm(args, pkg, world, rank, local_rank)
    if args.workload == "spectrum":
        return main_spectrum(args, pkg, world, rank, local_rank)
    fs = 2.4e6
    ns = args.streams_per_gpu
    nsamp = int(round(args.seconds * fs))
    nbytes = 2 * nsamp
    h, g = pkg.default_config(args.fir_taps, fs=fs)
    D, Da = 10, 5

    # synthetic input: FM test signal, distinct PRNG stream per (rank, stream)
    distinct = args.distinct if args.distinct > 0 else ns
    t_gen = time.perf_counter()
    base = pkg.make_iq(distinct, nsamp, mode="fm", fs=fs, first_id=rank * ns)
    iq_host = base if distinct == ns else np.tile(base, ((ns + distinct - 1) // distinct, 1))[:ns]
    t_gen = time.perf_counter() - t_gen

    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, device=local_rank))
    stream = torch.cuda.Stream()
    dm.set_stream(stream.cuda_stream)
    n_audio_max = nsamp // D // Da + 1
    with torch.cuda.stream(stream):
        iq = torch.from_numpy(iq_host).cuda()
        audio = torch.zeros((ns, n_audio_max), dtype=torch.float32, device="cuda")
    stream.synchronize()

    e2e = None
    if args.end_to_end:
        # rank 0 owns the whole batch; each step = C1 scatter -> local hot path -> C2 gather, all on `stream`
        if not dist.is_initialized():
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        fan = pkg.fanout
        with torch.cuda.stream(stream):
            iq_all = torch.from_numpy(np.tile(iq_host, (world, 1))).cuda() if rank == 0 else None
        stream.synchronize()
        e2e = {"iq_all": iq_all, "total": world * ns}

    def step():
        if e2e is None:
            return dm.process_batch_device(iq, audio)
        with torch.cuda.stream(stream):
            local = pkg.fanout.scatter_streams(e2e["iq_all"], e2e["total"], nbytes, iq.device)
            n = dm.process_batch_device(local, audio)
            pkg.fanout.gather_audio(audio[:, :n], e2e["total"])
        return n

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Timed region: K back-to-back launches between ONE pair of HIP events on the launch stream (an event pair around every
    # launch costs ~6 us of idle GPU per step, 14 % of this kernel).  kernel_ms_avg = event span / K is therefore the average
    # launch duration INCLUDING any inter-launch gap; per-launch event timings come from a short untimed pass afterwards.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    n_audio = 0
    ev0.record(stream)
    for _ in range(args.steps):
        n_audio = step()
    ev1.record(stream)
    fence()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms_avg = ev0.elapsed_time(ev1) / args.steps
    kernel_ms = per_launch_events(torch, stream, step, min(args.steps, 20))
    fence()

    ok = None
    if args.check and rank == 0:
        from oracle.oracle import Oracle
        dm.reset()
        step()
        dm.synchronize()
        got = audio[:, :n_audio].cpu().numpy()
        ok = True
        for s in (0, ns // 2, ns - 1):
            want = Oracle(h, g).process(iq_host[s])
            err = np.max(np.abs(got[s] - want) / np.maximum(np.abs(want), 1.0))
            ok = ok and bool(err <= 1e-5)

    if rank == 0:
        total_samples = float(world) * ns * nsamp * args.steps
        value = total_samples / elapsed / 1e6
        samples_per_launch = ns * nsamp
        alg_bytes = samples_per_launch * 2.0 + ns * n_audio * 4.0        # 2 B in + 4/(D*Da) B out per IQ sample
        achieved = alg_bytes / (kernel_ms_avg * 1e-3) / 1e9
        traffic = latest_traffic(dm.kernel_name)
        res = {
            "metric": "IQ MSamples/s through FIR+FM-demod+resample",
            "value": round(value, 1), "unit": "MSamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "mode": "end-to-end (RCCL scatter of IQ from rank 0 + gather of audio every step)" if args.end_to_end else "compute-only (IQ resident per GPU)",
            "config": {"workload": "BASELINE configs[2]: %d concurrent 2.4 MS/s uint8 IQ streams per GPU x %.1f s (%d B each), "
                                   "%d-tap FIR /%d + FM discriminator + %d-tap /%d -> 48 kHz; device-resident, streams sharded "
                                   "across GPUs with no collective" % (ns, args.seconds, nbytes, args.fir_taps, D, len(g), Da),
                       "streams_per_gpu": ns, "bytes_per_stream": nbytes, "fir_taps": args.fir_taps, "kernel": dm.kernel_name,
                       "bytes_per_sample_algorithmic": round(alg_bytes / samples_per_launch, 4),
                       "GB_per_s_input": round(value * 2e6 / 1e9, 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                         "kernel": dm.kernel_name, "kernel_ms_avg": round(kernel_ms_avg, 4),
                         "kernel_ms_isolated_avg": round(float(np.mean(kernel_ms)), 4),
                         "kernel_ms_min": round(float(np.min(kernel_ms)), 4),
                         "algorithmic_bytes_per_launch": alg_bytes},
            "gen_seconds": round(t_gen, 2),
        }
        if ok is not None:
            res["parity_ok"] = ok
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(pkg, h, g, iq_host[: min(ns, 64)], args.cpu_seconds, os.cpu_count() or 1)
        print(json.dumps(res), flush=True)
    dm.set_stream(None)
    dm.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def per_launch_events(torch, stream, step, n):
    """Untimed pass: one HIP event pair around each of n launches on the launch stream -> list of ms."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


def main_wbfm(args, pkg, world, rank, local_rank):
    """BASELINE configs[4]: 3.2 MS/s IQ, 128-tap prototype, 16 bands, per-band FM demod, 6/25 resampler; 128 streams/GPU."""
    import numpy as np
    import torch
    import torch.distributed as dist
    fs, ns = 3.2e6, (args.streams_per_gpu if args.streams_per_gpu != 256 else 128)
    nsamp = int(round(args.seconds * fs))
    p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8)
    g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
    iq_host = pkg.make_iq(ns, nsamp, mode="fm", fs=fs, first_id=rank * ns)
    dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, device=local_rank, max_bytes_per_call=2 * nsamp))
    kname = dm.kernel_name
    stream = torch.cuda.Stream()
    dm.set_stream(stream.cuda_stream)
    cap = dm.audio_count(2 * nsamp) + 8
    with torch.cuda.stream(stream):
        iq = torch.from_numpy(iq_host).cuda()
        audio = torch.zeros((ns, 16, cap), dtype=torch.float32, device="cuda")
    stream.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        dm.process_batch_device(iq, audio)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    n_audio = 0
    ev0.record(stream)
    for _ in range(args.steps):
        n_audio = dm.process_batch_device(iq, audio)
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = ev0.elapsed_time(ev1) / args.steps                      # launch duration incl. inter-launch gap (see main())
    if rank == 0:
        alg = ns * nsamp * 2.0 + ns * 16 * n_audio * 4.0
        res = {"metric": "IQ MSamples/s through FIR+FM-demod+resample", "value": round(world * ns * nsamp * args.steps / elapsed / 1e6, 1),
               "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[4]: %d x 3.2 MS/s uint8 IQ streams per GPU x %.1f s, 128-tap prototype, 16-band polyphase "
                                      "channelizer + per-band FM demod + 6/25 resampler -> 16 x 48 kHz" % (ns, args.seconds),
                          "streams_per_gpu": ns, "bytes_per_stream": 2 * nsamp, "kernel": kname},
               "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "traffic": (latest_traffic(kname) or {}).get("hbm_bytes_per_launch"), "kernel_ms_avg": round(ms, 4),
                            "algorithmic_bytes_per_launch": alg}}
        print(json.dumps(res), flush=True)
    dm.set_stream(None)
    dm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_spectrum(args, pkg, world, rank, local_rank):
    """FFT view of the BASELINE configs[2] buffers: 256 x 0.1 s of 2.4 MS/s IQ per GPU -> averaged power spectra."""
    import numpy as np
    import torch
    import torch.distributed as dist
    fs, ns, nfft = 2.4e6, args.streams_per_gpu, args.nfft
    nsamp = int(round(args.seconds * fs))
    iq_host = pkg.make_iq(ns, nsamp, mode="fm", fs=fs, first_id=rank * ns)
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, device=local_rank, max_bytes_per_call=2 * nsamp))
    stream = torch.cuda.Stream()
    sv.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        iq = torch.from_numpy(iq_host).cuda()
        power = torch.zeros((ns, nfft), dtype=torch.float32, device="cuda")
    stream.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sv.process_batch_device(iq, power)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    frames = 0
    ev0.record(stream)
    for _ in range(args.steps):
        frames = sv.process_batch_device(iq, power)
    ev1.record(stream)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = ev0.elapsed_time(ev1) / args.steps                      # launch duration incl. inter-launch gap (see main())
    if rank == 0:
        alg = ns * frames * nfft * 2.0 + ns * nfft * 4.0
        res = {"metric": "IQ MSamples/s through the windowed-FFT spectrum view", "value": round(world * ns * frames * nfft * args.steps / elapsed / 1e6, 1),
               "unit": "MSamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "spectrum view (SURVEY 8f-3) of BASELINE configs[2] buffers: %d x 2.4 MS/s uint8 IQ streams per GPU x %.1f s, "
                                      "%d-point Hann FFT, %d frames averaged per stream" % (ns, args.seconds, nfft, frames),
                          "streams_per_gpu": ns, "bytes_per_stream": 2 * nsamp, "kernel": "k_spectrum<%d>" % int(np.log2(nfft))},
               "roofline": {"bound": "hbm", "achieved": round(alg / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "traffic": (latest_traffic("k_spectrum<%d>" % int(np.log2(nfft))) or {}).get("hbm_bytes_per_launch"),
                            "kernel_ms_avg": round(ms, 4), "algorithmic_bytes_per_launch": alg}}
        print(json.dumps(res), flush=True)
    sv.set_stream(None)
    sv.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
