#!/usr/bin/env python3
"""Replay a case dumped by tools/fuzz_q.py (FUZZ_DUMP=dir): same rows, sizes and resets, fast handle vs bit-exact handle, verbose."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
for path in sys.argv[1:]:
    z = np.load(path)
    h, g, ns, stride, rows, sizes = z["h"], z["g"], int(z["ns"]), int(z["stride"]), z["rows"], [int(x) for x in z["sizes"]]
    # `resets` holds, for each reset, its index in the sequence [reset | call] entries: turn into "reset before call k"
    seq, k, before = list(z["resets"]), 0, set()
    pos_in_seq = 0
    for ci in range(len(sizes)):
        while pos_in_seq in seq:
            before.add(ci); pos_in_seq += 1
        pos_in_seq += 1
    nd, total = rows.shape[0], rows.shape[1] // 2
    dev = torch.zeros((ns, stride), dtype=torch.uint8, device="cuda")
    dev[:, :2 * total] = torch.from_numpy(np.tile(rows, ((ns + nd - 1) // nd, 1))[:ns]).cuda()
    torch.cuda.synchronize()
    kw = dict(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * max(sizes) + 64)
    fast = pkg.FmDemod(pkg.FmConfig(**kw)); exact = pkg.FmDemod(pkg.FmConfig(bit_exact=True, **kw))
    pos = 0
    print(path, "T", len(h), "ns", ns, "sizes", sizes, "resets before calls", sorted(before))
    for ci, n in enumerate(sizes):
        if ci in before:
            fast.reset(); exact.reset()
        cap = fast.audio_count(2 * n) + 1
        a1 = torch.full((ns, cap), 3.0, dtype=torch.float32, device="cuda"); a2 = torch.full((ns, cap), 5.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        n1 = fast.process_batch_device(dev[:, 2 * pos:], a1, nbytes=2 * n); name = fast.kernel_name
        n2 = exact.process_batch_device(dev[:, 2 * pos:], a2, nbytes=2 * n)
        fast.synchronize(); exact.synchronize()
        g1, g2 = a1[:, :n1].cpu().numpy().astype(np.float64), a2[:, :n2].cpu().numpy().astype(np.float64)
        if n1:
            e = np.abs(g1 - g2) / np.maximum(np.abs(g2), 1.0)
            s, j = np.unravel_index(np.argmax(e), e.shape)
            bad = np.argwhere(e > 1e-5)
            print("  call", ci, n, name, "max err %.3g at stream %d output %d of %d" % (e.max(), s, j, n1), "got", g1[s, j], "want", g2[s, j],
                  "| bad outputs %d, streams %s, outputs %s" % (len(bad), sorted(set(bad[:, 0] % nd))[:8], sorted(set(bad[:, 1]))[:12]))
        pos += n
    fast.close(); exact.close()
