#!/usr/bin/env python3
"""Repro of a fuzz_q failure: T=64, 64 streams, call of 16400 samples, reset, call of 15200 samples (first call after a reset on design Q)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle.oracle import Oracle
h, g = pkg.default_config(64)
ns, sizes = 64, [16400, 15200]
total = sum(sizes)
for mode in ("random", "const", "counter"):
    for extra in (0, 16, 48, 2, 6):
        rows = np.concatenate([pkg.make_iq(4, total, mode="fm", first_id=11), pkg.make_iq(2, total, mode=mode, first_id=77)])
        stride = 2 * total + extra
        dev = torch.zeros((ns, stride), dtype=torch.uint8, device="cuda")
        dev[:, :2 * total] = torch.from_numpy(np.tile(rows, (11, 1))[:ns]).cuda()
        torch.cuda.synchronize()
        kw = dict(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * max(sizes) + 64)
        fast = pkg.FmDemod(pkg.FmConfig(**kw)); exact = pkg.FmDemod(pkg.FmConfig(bit_exact=True, **kw))
        pos = 0
        for i, n in enumerate(sizes):
            if i == 1:
                fast.reset(); exact.reset()
            cap = fast.audio_count(2 * n) + 1
            a1 = torch.full((ns, cap), 3.0, dtype=torch.float32, device="cuda"); a2 = torch.full((ns, cap), 5.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            n1 = fast.process_batch_device(dev[:, 2 * pos:], a1, nbytes=2 * n); name = fast.kernel_name
            n2 = exact.process_batch_device(dev[:, 2 * pos:], a2, nbytes=2 * n)
            fast.synchronize(); exact.synchronize()
            g1, g2 = a1[:, :n1].cpu().numpy().astype(np.float64), a2[:, :n2].cpu().numpy().astype(np.float64)
            e = np.abs(g1 - g2) / np.maximum(np.abs(g2), 1.0)
            s, j = np.unravel_index(np.argmax(e), e.shape)
            if e.max() > 1e-5:
                bad = np.argwhere(e > 1e-5)
                print(mode, extra, "call", i, name, "max err %.3g at stream %d output %d; bad outputs: %d, streams %s, outputs %d..%d" % (e.max(), s, j, len(bad), sorted(set(bad[:, 0]))[:8], bad[:, 1].min(), bad[:, 1].max()), "got", g1[s, j], "want", g2[s, j])
            pos += n
        fast.close(); exact.close()
print("done")
