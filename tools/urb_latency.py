#!/usr/bin/env python3
"""Per-call latency of the synchronous single-stream hand-off (sdrfm_process) for URB-sized chunks, and the real-time margin
at 2.4 MS/s (a 512-byte URB carries 106.7 us of signal; the reference's buffSize range is 512 .. 127*512 bytes)."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
h, g = pkg.default_config(64)
iq = pkg.make_iq(1, 1 << 21, mode="fm")[0]
res = []
for nbytes in (512, 4096, 16384, 127 * 512, 262144):
    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, max_bytes_per_call=262144))
    calls = min(2000, iq.size // nbytes)
    for i in range(20):
        dm.process(iq[i * nbytes:(i + 1) * nbytes])
    t0 = time.perf_counter()
    for i in range(calls):
        dm.process(iq[i * nbytes:(i + 1) * nbytes])
    dt = (time.perf_counter() - t0) / calls
    res.append({"urb_bytes": nbytes, "us_per_call": round(dt * 1e6, 1), "signal_us_per_urb": round(nbytes / 2 / 2.4, 1),
                "realtime_factor": round((nbytes / 2 / 2.4e6) / dt, 2), "kernel": dm.kernel_name})
    dm.close()
print(json.dumps(res))
