#!/usr/bin/env python3
"""Randomised soak of the design-S kernel (streaming lanes): machine-filling batches, call sizes that are / are not whole lane
segments, resets in mid-stream, device-resident buffers with odd row strides.  Every call is compared BIT FOR BIT with the
generic kernel on a twin handle.  usage: fuzz_stream.py [seconds] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end, cases, calls_s, fails = time.time() + budget, 0, 0, 0
while time.time() < t_end:
    T = int(rng.choice([64, 32]))                               # both instantiated design-S geometries (the 32-tap one has an odd number of head outputs)
    h, g = pkg.default_config(T)
    ns = int(rng.choice([205, 256, 300, 512, 1024, 1500]))
    sizes = []
    for _ in range(int(rng.integers(2, 6))):
        k = int(rng.integers(1, 12))
        sizes.append(2400 * k if rng.random() < 0.75 else int(rng.integers(1, 30000)))
    total = sum(sizes)
    nd = 8
    rows = np.concatenate([pkg.make_iq(nd - 2, total, mode="fm", first_id=int(rng.integers(1 << 20))),
                           pkg.make_iq(2, total, mode=str(rng.choice(["random", "const", "counter"])), first_id=int(rng.integers(1 << 20)))])
    stride = 2 * total + int(rng.choice([0, 16, 48, 2, 6]))          # strides that are / are not multiples of 16
    dev = torch.zeros((ns, stride), dtype=torch.uint8, device="cuda")
    dev[:, :2 * total] = torch.from_numpy(np.tile(rows, ((ns + nd - 1) // nd, 1))[:ns]).cuda()
    torch.cuda.synchronize()
    kw = dict(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * max(sizes) + 64)
    fast = pkg.FmDemod(pkg.FmConfig(bit_exact=True, **kw)); gen = pkg.FmDemod(pkg.FmConfig(force_generic=True, **kw))   # bit_exact: design S, not Q
    pos, log, bad = 0, [], False
    for n in sizes:
        if rng.random() < 0.15:
            fast.reset(); gen.reset(); log.append("reset")
        cap = fast.audio_count(2 * n) + 1
        a1 = torch.full((ns, cap), 3.0, dtype=torch.float32, device="cuda"); a2 = torch.full((ns, cap), 5.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        n1 = fast.process_batch_device(dev[:, 2 * pos:], a1, nbytes=2 * n); name = fast.kernel_name.split()[0]
        n2 = gen.process_batch_device(dev[:, 2 * pos:], a2, nbytes=2 * n)
        fast.synchronize(); gen.synchronize()
        log.append((n, name, T)); calls_s += name == "fast-s"
        if name == "fast-s" and ("T%d " % T) not in fast.kernel_name:
            bad = True
        if n1 != n2 or not torch.equal(a1[:, :n1].view(torch.int32), a2[:, :n2].view(torch.int32)):
            bad = True
        pos += n
    fast.close(); gen.close()
    cases += 1
    if bad:
        fails += 1; print("FAIL", dict(ns=ns, stride=stride, log=log), flush=True)
print("design-S soak: cases %d  calls served by fast-s %d  failures %d  (seed %d, %.0f s)" % (cases, calls_s, fails, seed, budget))
sys.exit(1 if fails else 0)
