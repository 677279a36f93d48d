#!/usr/bin/env python3
"""Host-buffer (PCIe-inclusive) rate of sdrfm_process_batch on the bench workload — reported in DESIGN.md, never as `value`."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns, nsamp = 256, 240000
h, g = pkg.default_config(64)
iq = np.tile(pkg.make_iq(32, nsamp), (8, 1))
dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp))
for _ in range(2):
    dm.process_batch(iq)
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    a = dm.process_batch(iq)
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"what": "sdrfm_process_batch, pageable host buffers in/out (H2D + kernel + D2H, synchronous)",
                  "streams": ns, "bytes_per_stream": 2 * nsamp, "ms_per_call": round(dt * 1e3, 3),
                  "MSamples_per_s": round(ns * nsamp / dt / 1e6, 1), "GB_per_s_in": round(ns * 2 * nsamp / dt / 1e9, 2),
                  "kernel": dm.kernel_name}))
