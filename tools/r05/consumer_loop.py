"""tools/r05/consumer_loop.py — what the consumer loop of INTEGRATION.md costs per call: overlapped calls with sdrfm_flush_previous after every call (the handle's stream
ordered behind call k-1, where a consumer of its audio would run) against the same calls flushed once at the end.  configs[2] shape, 5 rotated input batches, regions of 300 calls,
the median of the last five of ten regions.  Measurement only."""
import sys, os, time, importlib
import numpy as np
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
import torch
ns, nsamp = 256, 240000
h, g = pkg.default_config(64)
iq = pkg.make_iq(16, nsamp * 5, mode="fm", first_id=1)
batches = [torch.from_numpy(np.tile(iq[:, 2 * k * nsamp:2 * (k + 1) * nsamp], (ns // 16, 1))).cuda() for k in range(5)]
aud = [torch.zeros((ns, nsamp // 50), dtype=torch.float32, device="cuda") for _ in range(3)]
torch.cuda.synchronize()
st = torch.cuda.Stream()
def region(dm, n, per_call):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for k in range(n):
        dm.process_batch_device(batches[k % 5], aud[k % 3], overlap=True)
        if per_call and k:
            dm.flush(keep_last=True)
    dm.flush()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp)) as dm:
    dm.set_stream(st.cuda_stream)
    for per_call in (False, True, False, True):
        time.sleep(0.3)
        r = [region(dm, 300, per_call) for _ in range(10)]
        print("flush_previous after every call: %-5s  us per call, ten regions of 300: %s   steady (median of the last five) %.2f" % (per_call, " ".join("%.2f" % x for x in r), float(np.median(r[5:]))))
