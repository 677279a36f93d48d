#!/usr/bin/env python3
"""Per-wave phase timeline of the WBFM step kernel (k_wbfm_steps) on the configs[4] shape.  Runs the DEVELOPMENT library
(csrc/libsdrfm_dev.so) with SDRFM_WBFM_PROFILE=1; a stamp waits for everything outstanding, so each phase includes its own
memory latency and the kernel runs slower than the product build (diagnostic, not a benchmark).

    python tools/wbfm_profile.py            # 128 streams: two waves per SIMD
    NS=64 RS=1250 python tools/wbfm_profile.py    # 64 streams, 1250-step runs: ONE wave per SIMD (a wave's own critical path)
"""
import ctypes as C, importlib, os, sys
LIGHT = os.environ.get("LIGHT", "0") == "1"             # LIGHT=1: only entry / exit times, the kernel runs at full speed
os.environ["SDRFM_WBFM_PROFILE"] = "2" if LIGHT else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns, nsamp = int(os.environ.get("NS", "128")), 320000
p = pkg.lowpass_taps(128, 0.5 / 16 * 0.8); g = pkg.lowpass_taps(60, 0.5 / 25 * 0.8) * 6.0
base = torch.from_numpy(pkg.make_iq(64, nsamp, mode="fm", fs=3.2e6)).cuda()
batches = [torch.cat([torch.roll(base, shifts=2 * (7919 * (b * 2 + r) % nsamp), dims=1) for r in range((ns + 63) // 64)])[:ns].contiguous() for b in range(5)]
dm = pkg.WbfmDemod(pkg.WbfmConfig(proto_coeffs=p, resamp_coeffs=g, n_streams=ns, max_bytes_per_call=2 * nsamp,
                                  run_steps=int(os.environ.get("RS", "0")), dev_library=True))
assert "k_wbfm_steps" in dm.kernel_name, dm.kernel_name
audio = torch.zeros((ns, 16, dm.audio_count(2 * nsamp) + 8), dtype=torch.float32, device="cuda")
for i in range(8):
    dm.process_batch_device(batches[i % 5], audio)
dm.synchronize()
lib = pkg.load_library(dev=True)
raw = (C.c_uint64 * (256 * 16384))()
nw = C.c_uint32()
lib.sdrfm_wbfm_dev_read_debug.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint32)]
assert lib.sdrfm_wbfm_dev_read_debug(dm._h, raw, 16384, C.byref(nw)) == 0
t = np.frombuffer(raw, dtype=np.uint64).reshape(-1, 256)[:nw.value].astype(np.int64)
NP = 7
if LIGHT:
    t[:, 2:250] = 0
nb = int(((t[:, 2:250] != 0).sum(axis=1).min()) // NP)        # whole blocks every wave ran
d = np.diff(t[:, :2 + NP * nb], axis=1)
ph = d[:, 1:1 + NP * nb].reshape(-1, nb, NP) if nb else np.zeros((len(t), 0, NP))
names = ["block top (tap loads)", "FIR", "next tile store (+ prefetch wait)", "resampler reads + DFT", "resampler chains + stores", "discriminator", "d rows, bookkeeping"]
print("kernel:", dm.kernel_name, " waves:", nw.value, " blocks per wave:", nb, "" if LIGHT else " prologue: %.0f cycles" % d[:, 0].mean())
print("%-36s %10s %10s %10s" % ("phase (cycles, blocks 1..n-2)", "mean", "p10", "p90"))
mid = ph[:, 1:-1] if nb > 2 else ph
for i, n in enumerate(names if nb else []):
    print("%-36s %10.0f %10.0f %10.0f" % (n, mid[:, :, i].mean(), np.percentile(mid[:, :, i], 10), np.percentile(mid[:, :, i], 90)))
if nb:
    print("%-36s %10.0f" % ("block", mid.sum(axis=2).mean()))
rt0, rt1, slot = t[:, 254], t[:, 255], t[:, 253] & 0xf
print("wave duration us: mean %.1f  p10 %.1f  p90 %.1f   (waves per SIMD slot: %s)" % (((rt1 - rt0) / 100).mean(), np.percentile((rt1 - rt0) / 100, 10), np.percentile((rt1 - rt0) / 100, 90),
      dict(zip(*[x.tolist() for x in np.unique(slot, return_counts=True)]))))
for sl in np.unique(slot):
    m = slot == sl
    print("slot %d: ends p50 %.1f us after the first start" % (sl, (np.median(rt1[m]) - rt0.min()) / 100))
