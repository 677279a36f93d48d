import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
ns, nsamp = int(os.environ.get("NS", "128")), 320000
dbg = torch.zeros(256 * 4096, dtype=torch.int64, device="cuda")
os.environ["WDBG"] = str(dbg.data_ptr())
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8); g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
base = torch.from_numpy(pkg.make_iq(64, nsamp, mode="fm", fs=3.2e6)).cuda()
batches = [torch.cat([torch.roll(base, shifts=2 * (7919 * (b * 2 + r) % nsamp), dims=1) for r in range((ns + 63) // 64)])[:ns].contiguous() for b in range(5)]
dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp, run_steps=int(os.environ.get('RS', '0'))))
audio = torch.zeros((ns, 16, dm.audio_count(2 * nsamp) + 8), dtype=torch.float32, device="cuda")
for i in range(8): dm.process_batch_device(batches[i % 5], audio)
dm.synchronize(); torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(-1, 256)[:int(os.environ.get('NW', '2048'))]
nb = 20
NP = 7
cyc = t[:, :2 + NP * nb]
d = np.diff(cyc, axis=1)
print("prologue cycles: mean %.0f" % d[:, 0].mean())
ph = d[:, 1:1 + NP * nb].reshape(-1, nb, NP)
names = ["top (taps issue)", "FIR", "next tile store", "resample issue + DFT", "resample finish", "disc", "d rows"]
for i, n in enumerate(names):
    print("%-16s mean %.0f  p10 %.0f p90 %.0f   (first block %.0f, last %.0f)" % (n, ph[:, 1:-1, i].mean(), np.percentile(ph[:, 1:-1, i], 10), np.percentile(ph[:, 1:-1, i], 90), ph[:, 0, i].mean(), ph[:, -1, i].mean()))
print("block mean %.0f" % ph[:, 1:-1].sum(axis=2).mean())
rt0, rt1 = t[:, 254], t[:, 255]
slot = t[:, 253] & 0xf
print("starts spread us: %.2f .. %.2f ; ends: p0 %.2f p10 %.2f p50 %.2f p90 %.2f p100 %.2f" % (0, (rt0.max() - rt0.min()) / 100, *[(np.percentile(rt1, q) - rt0.min()) / 100 for q in (0, 10, 50, 90, 100)]))
for sl in np.unique(slot):
    m = slot == sl
    print("slot", sl, "n", m.sum(), "start p50 %.2f end p50 %.2f us" % ((np.median(rt0[m]) - rt0.min()) / 100, (np.median(rt1[m]) - rt0.min()) / 100))
