#!/usr/bin/env python3
"""One very large device-resident call (hundreds of MB per stream) against the oracle: 32-bit index arithmetic at scale."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle import oracle as om
for T, nsamp, ns in ((64, 400_000_010, 1), (16, 150_000_000, 2)):
    h, g = pkg.default_config(T)
    iq_host = pkg.make_iq(ns, nsamp, mode="random", first_id=3)
    dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp))
    iq = torch.from_numpy(iq_host).cuda()
    audio = torch.zeros((ns, dm.audio_count(2 * nsamp) + 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = dm.process_batch_device(iq, audio); dm.synchronize(); dt = time.perf_counter() - t0
    got = audio[:, :n].cpu().numpy()
    worst = 0.0
    for s in range(ns):
        want = om.Oracle(h, g).process(iq_host[s])
        assert want.size == n, (want.size, n)
        worst = max(worst, float(np.max(np.abs(got[s] - want) / np.maximum(np.abs(want), 1.0))))
    print("T=%d ns=%d samples/stream=%d kernel=%s  %.1f ms  worst scaled error %.3g" % (T, ns, nsamp, dm.kernel_name, dt * 1e3, worst))
    assert worst <= 1e-5
    dm.close(); del iq, audio
print("ok")
