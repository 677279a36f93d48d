import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
from oracle.oracle import Oracle
T, ns, calls = 64, 300, [24000, 1000, 24000, 2402, 2398, 48000]
h, g = pkg.default_config(T)
total = sum(calls)
rows = pkg.make_iq(4, total, mode="fm", first_id=700)
iq = np.tile(rows, (ns // 4, 1))[:ns]
kw = dict(fir_coeffs=h, audio_coeffs=g, n_streams=ns, max_bytes_per_call=2 * max(calls))
fast = pkg.FmDemod(pkg.FmConfig(**kw)); exact = pkg.FmDemod(pkg.FmConfig(bit_exact=True, **kw))
o = Oracle(h, g)
pos = 0
for n in calls:
    a = fast.process_batch(iq[:, 2 * pos:2 * (pos + n)]); name = fast.kernel_name
    b = exact.process_batch(iq[:, 2 * pos:2 * (pos + n)]); nb = exact.kernel_name
    w = o.process(iq[0, 2 * pos:2 * (pos + n)])
    e = np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b), 1)
    eo = np.abs(a[0].astype(np.float64) - w) / np.maximum(np.abs(w), 1)
    bad = np.argwhere(e > 2e-6)
    print(n, name, "|", nb, "| vs exact max %.3g nbad %d first %s | vs oracle row0 %.3g first bad idx %s" % (e.max(), len(bad), bad[:3].tolist(), eo.max(), np.argwhere(eo > 1e-5)[:5].ravel().tolist()))
    pos += n
