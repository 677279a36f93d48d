"""tools/consumer_loop.py — what the consumer loop of INTEGRATION.md costs per call: overlapped calls with sdrfm_flush_previous after every call (the handle's stream
ordered behind call k-1, where a consumer of its audio runs) against the same calls flushed once at the end; then the same loop WITH the device PCM sink as that consumer,
in its default form (the blocked scan) and in its exact form (one lane per stream), and the sink alone.  configs[2] shape, 5 rotated input batches, regions of 300 calls,
the median of the last five of ten regions.  Measurement only (profiles/r06_sink.txt)."""
import sys, os, time, importlib
if len(sys.argv) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]   # e.g. 8: before the HIP runtime loads (the loop uses five streams; the runtime's default pool holds 4 hardware queues)
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
import torch
ns, nsamp = 256, 240000
h, g = pkg.default_config(64)
iq = pkg.make_iq(16, nsamp * 5, mode="fm", first_id=1)
batches = [torch.from_numpy(np.tile(iq[:, 2 * k * nsamp:2 * (k + 1) * nsamp], (ns // 16, 1))).cuda() for k in range(5)]
aud = [torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda") for _ in range(6)]
NA = int(os.environ.get("LOOP_AUDIO_BUFFERS", "3"))   # audio buffers of the fast form
torch.cuda.synchronize()
st = torch.cuda.Stream()
pcm = [torch.zeros((ns, 2 * (nsamp // 50)), dtype=torch.int16, device="cuda") for _ in range(2)]
lib = pkg.load_library()
alpha, gain = lib.sdrfm_pcm_alpha(48000.0, 75e-6), float(np.float32(32767.0 / (2 * np.pi * 75e3 / 240e3)))
def region(dm, n, per_call, sink=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for k in range(n):
        dm.process_batch_device(batches[k % 5], aud[k % 3], overlap=True)
        if per_call and k:
            dm.flush(keep_last=True)
            if sink is not None:
                sink.process_batch_device(aud[(k - 1) % 3], pcm[k & 1], nsamp // 50)
    dm.flush()
    if sink is not None:
        sink.process_batch_device(aud[(n - 1) % 3], pcm[n & 1], nsamp // 50)
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ONLY_FAST = bool(os.environ.get("LOOP_ONLY_FAST"))            # (for a kernel trace of the fast form alone)
with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
    dm.set_stream(st.cuda_stream)
    for per_call in (() if ONLY_FAST else (False, True, False, True)):
        time.sleep(0.3)
        r = [region(dm, 300, per_call) for _ in range(10)]
        print("flush_previous after every call: %-5s  us per call, ten regions of 300: %s   steady (median of the last five) %.2f" % (per_call, " ".join("%.2f" % x for x in r), float(np.median(r[len(r) // 2:]))))
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES", "(default: 4)"))
    for exact in (() if ONLY_FAST else (False,) if len(sys.argv) > 2 else (False, True, False)):
        with pkg.PcmSink(ns, alpha, gain, exact=exact) as sink:
            sink.set_stream(st.cuda_stream)
            time.sleep(0.3)
            nreg = 3 if exact else 10
            r = [region(dm, 60 if exact else 300, True, sink) for _ in range(nreg)]
            print("simple form (two audio buffers, the device PCM sink on the handle's stream, %s): us per call: %s   steady %.2f" % ("exact: one lane per stream" if exact else "default: blocked scan", " ".join("%.2f" % x for x in r), float(np.median(r[nreg // 2:]))))
            # the sink alone: launches back to back
            nl = 30 if exact else 300
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.synchronize(); e0.record(st)
            for k in range(nl):
                sink.process_batch_device(aud[k % 3], pcm[k & 1], nsamp // 50)
            e1.record(st); e1.synchronize()
            print("   the sink alone (%s): %.2f us per 256 x 4800 launch" % ("exact" if exact else "scan", e0.elapsed_time(e1) / nl * 1e3))
    # the fast form of INTEGRATION.md: the sink on a stream of its own behind sdrfm_wait_previous, three audio buffers — only call k + 2 has the consumer of call k - 1 to mind
    sst = torch.cuda.Stream()
    for rep in range(2):
        with pkg.PcmSink(ns, alpha, gain) as sink:
            sink.set_stream(sst.cuda_stream)
            consumed = [None] * NA

            def region3(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for k in range(n):
                    if consumed[k % NA] is not None:
                        while not consumed[k % NA].query():         # audio[k % NA] is free once its consumer (of call k - NA) is done: the HOST waits for that (a busy
                            pass                                    # poll: it stays at most NA calls ahead of the device); no wait enters a queue of the demodulator's
                    dm.process_batch_device(batches[k % 5], aud[k % NA], overlap=True)
                    if k:
                        dm.wait_previous(sst.cuda_stream)          # the sink's stream behind call k - 1; the handle's stream stays idle
                        sink.process_batch_device(aud[(k - 1) % NA], pcm[(k - 1) & 1], nsamp // 50)
                        consumed[(k - 1) % NA] = torch.cuda.Event(); consumed[(k - 1) % NA].record(sst)
                dm.flush()
                st.wait_stream(sst)
                e1.record(st)
                e1.synchronize()
                return e0.elapsed_time(e1) / n * 1e3
            time.sleep(0.3)
            r = [region3(300) for _ in range(3 if ONLY_FAST else 10)]
            print("fast form (%d audio buffers, the sink on its own stream, the host at most that many calls ahead): us per call" % NA + ", ten regions of 300: %s   steady %.2f" % (" ".join("%.2f" % x for x in r), float(np.median(r[len(r) // 2:]))))
