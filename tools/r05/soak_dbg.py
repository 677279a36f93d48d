"""tools/r05/soak_dbg.py <seed> — test infrastructure (the oracle is the checker here, as in tests/): the routing soak of tests/test_route_gpu.py, call by call: which streams of which call miss the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle import oracle as oracle_mod
import torch
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
T, D, Da = [(64, 10, 5), (16, 10, 5), (64, 8, 8), (64, 16, 5), (32, 10, 5), (64, 10, 5)][seed % 6]
h, g = pkg.default_config(T, fir_decim=D, audio_taps=32, audio_decim=Da)
ns = int(rng.choice([96, 160, 256]))
unit = D * Da * 8
lens = [int(unit * rng.integers(60, 220)) for _ in range(14)]
total = sum(lens)
fm = pkg.make_iq(6, total, mode="fm", first_id=5000 + seed)
rnd = pkg.make_iq(3, total, mode="random", first_id=5100 + seed)
kind = rng.integers(0, 9, size=ns)
group = rng.integers(0, 3, size=ns)
rows = [fm[k] if k < 6 else rnd[k - 6] for k in range(9)]
want = [oracle_mod.Oracle(h, g, D=D, Da=Da).process(r) for r in rows]
iq = np.stack([rows[k] for k in kind])
dev = torch.from_numpy(iq).cuda()
na_max = max(lens) // (D * Da)
with pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, fir_decim=D, audio_decim=Da, n_streams=ns, max_bytes_per_call=2 * max(lens))) as dm:
    off = 0; aoff = 0
    mask = np.zeros(ns, dtype=np.uint8)
    for k, n in enumerate(lens):
        if k and rng.random() < 0.5:
            on = rng.random(3) < [0.5, 0.35, 0.2]
            mask = on[group].astype(np.uint8)
            dm.route(mask)
        out = torch.zeros((ns, na_max), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ovl = bool(rng.random() < 0.6)
        na = dm.process_batch_device(dev[:, 2 * off:], out, nbytes=2 * n, overlap=ovl)
        name = dm.kernel_name
        dm.synchronize()
        o = out[:, :na].cpu().numpy()
        bad = []
        for s in range(ns):
            w = want[kind[s]][aoff:aoff + na]
            e = np.max(np.abs(o[s] - w) / np.maximum(np.abs(w), 1.0))
            if e > 1e-5:
                nz = int(np.count_nonzero(o[s])); first = int(np.argmax(np.abs(o[s] - w) / np.maximum(np.abs(w), 1.0) > 1e-5))
                bad.append((s, int(mask[s]), round(float(e), 6), nz, first))
        print("call %2d n=%6d na=%5d ovl=%d routed=%3d/%d  %-90s bad=%d %s" % (k, n, na, ovl, int(mask.sum()), ns, name, len(bad), bad[:6]))
        off += n; aoff += na
