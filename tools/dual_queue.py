#!/usr/bin/env python3
"""Experiment: the 256-stream batch as ONE handle on one HIP stream vs TWO 128-stream handles on two HIP streams (their
launches drift out of phase, so one half's ramp/tail overlaps the other half's steady state).  Prints us per 256-stream step."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
h, g = pkg.default_config(int(os.environ.get("TAPS", "64")))
nsamp = 240000
base = pkg.make_iq(8, nsamp)

def setup(ns):
    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp))
    s = torch.cuda.Stream(); dm.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        iq = torch.from_numpy(np.tile(base, (ns // 8, 1))).cuda()
        audio = torch.zeros((ns, 4808), dtype=torch.float32, device="cuda")
    s.synchronize()
    return dm, s, iq, audio

def run(parts, steps):
    for _ in range(10):
        for dm, s, iq, audio in parts: dm.process_batch_device(iq, audio)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        for dm, s, iq, audio in parts: dm.process_batch_device(iq, audio)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6

one = [setup(256)]
print("1 handle x 256 streams: %.1f us/step" % run(one, 300), "%.1f" % run(one, 300))
for k in (2, 4):
    parts = [setup(256 // k) for _ in range(k)]
    print("%d handles x %d streams: %.1f us/step" % (k, 256 // k, run(parts, 300)), "%.1f" % run(parts, 300))
