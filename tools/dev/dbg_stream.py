import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
h, g = pkg.default_config(64)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1
calls = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [240000]
total = sum(calls)
iq = pkg.make_iq(ns, total, mode="fm", first_id=900)
kw = dict(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * max(calls))
fast = pkg.FmDemod(pkg.FmConfig(**kw)); gen = pkg.FmDemod(pkg.FmConfig(force_generic=True, **kw))
pos = 0
for n in calls:
    a = fast.process_batch(iq[:, 2 * pos:2 * (pos + n)]); name = fast.kernel_name
    b = gen.process_batch(iq[:, 2 * pos:2 * (pos + n)]); pos += n
    bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
    print(n, name, "audio", a.shape, "mismatches", len(bad))
    if len(bad):
        js = np.unique(bad[:, 1])
        print("  streams", np.unique(bad[:, 0])[:10], "first j", js[:40], " ... last", js[-5:])
        print("  j*5 // 48 (lane-segment of newest d):", np.unique(js * 5 // 48)[:40])
        j = js[0]; print("  sample", a[bad[0][0], j], b[bad[0][0], j])
