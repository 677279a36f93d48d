#!/usr/bin/env python3
"""Determinism soak of the chained spectrum kernel (k_spectrum_chain): the running sum is handed from wave to wave through LDS behind a tag, which
relies on a wave's DS operations executing in order.  A violation would show as a launch whose output differs from the others on the same
input.  Repeats the bench-shape launch (256 x 234 x 1024, then 256 x 468 x 512, then ragged small shapes) many times and compares every
output with the first one bit for bit; the first is checked against the oracle for three streams.

    python tools/spectrum_soak.py [launches per shape]
"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle import oracle as om
n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
bad = 0
for ns, nfft, F in ((256, 1024, 234), (256, 512, 468), (256, 1024, 25), (64, 512, 49), (1024, 1024, 7), (256, 256, 937), (256, 64, 3750), (64, 128, 51)):
    rows = pkg.make_iq(8, nfft * F, mode="random" if F == 25 else "fm", first_id=1234 + F)
    iq_h = np.tile(rows, (ns // 8, 1))
    iq = torch.from_numpy(iq_h).cuda()
    ref = torch.zeros((ns, nfft), dtype=torch.float32, device="cuda")
    out = torch.zeros_like(ref)
    torch.cuda.synchronize()
    sv = pkg.SpectrumView(pkg.SpectrumConfig(nfft=nfft, n_streams=ns, max_bytes_per_call=2 * nfft * F))
    assert sv.process_batch_device(iq, ref) == F
    sv.synchronize()
    for s in (0, 3, ns - 1):
        want, _ = om.SpectrumOracle(nfft).process(iq_h[s])
        assert np.array_equal(ref[s].cpu().numpy().view(np.uint32), want.view(np.uint32)), (nfft, F, s)
    t0, diff = time.time(), 0
    for i in range(n_launch):
        sv.process_batch_device(iq, out)
        sv.synchronize()
        if not torch.equal(out, ref):
            diff += 1
    print("%s  %d streams x %d frames x %d points: %d launches, %d differ from the first  (%.1f s)" % (sv.kernel_name, ns, F, nfft, n_launch, diff, time.time() - t0), flush=True)
    bad += diff
    sv.close()
sys.exit(1 if bad else 0)
