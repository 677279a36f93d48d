#!/usr/bin/env python3
"""Phase stamps of the 512 / 1024-point spectrum kernel (k_spectrum_chain) on bench.py's spectrum shape (256 streams x 234 frames x 1024
points).  Runs the DEVELOPMENT library (csrc/libsdrfm_dev.so, built with -DSDRFM_DEV) with SDRFM_SPEC_STAMPS=1: every wave sums the shader
cycles (s_memtime) it spends per phase; the library prints the mean over the waves of the last launch when the handle is closed.  A stamp
waits for the wave's outstanding LDS operations, so the instrumented kernel runs ~5 % slower than the product's (diagnostic, not a benchmark).

    python tools/spectrum_stamps.py                 # the product's geometry: 12 waves per stream, runs of 2 blocks
    python tools/spectrum_stamps.py 81 82 121 123   # SDRFM_SPEC_VARIANT = waves x 10 + blocks per run
"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDRFM_SPEC_STAMPS"] = "1"
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns, nfft, F = 256, int(os.environ.get("NFFT", "1024")), 234
iq = torch.from_numpy(np.tile(pkg.make_iq(8, nfft * F, mode="fm", first_id=700), (ns // 8, 1))).cuda()
power = torch.zeros((ns, nfft), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for variant in (sys.argv[1:] or ["122"]):
    os.environ["SDRFM_SPEC_VARIANT"] = variant
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, max_bytes_per_call=2 * nfft * F, dev_library=True))
    for _ in range(5):
        sv.process_batch_device(iq, power)
    sv.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        sv.process_batch_device(iq, power)
    sv.synchronize()
    print("variant %s: %.1f us per launch (instrumented kernel, host clock over 50 launches)" % (variant, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
    sv.close()                                              # prints the stamps (stderr)
