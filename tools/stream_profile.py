#!/usr/bin/env python3
"""Per-wave timeline of the design-S kernel (k_stream) on the bench workload: where a wave's cycles go and how waves of one
launch line up in time.  Runs the DEVELOPMENT library (csrc/libsdrfm_dev.so) with SDRFM_STREAM_PROFILE=1."""
import ctypes as C, importlib, json, os, sys
os.environ["SDRFM_STREAM_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns = int(os.environ.get("STREAMS", "256")); nsamp = 240000; NB = 6
h, g = pkg.default_config(64)
dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, dev_library=True))
nb = int(os.environ.get("BATCHES", "5"))
base = torch.from_numpy(pkg.make_iq(64, nsamp)).cuda()
batches = [torch.cat([torch.roll(base, shifts=2 * (7919 * (b * 8 + r) % nsamp), dims=1) for r in range((ns + 63) // 64)])[:ns].contiguous() for b in range(nb)]
audio = torch.zeros((ns, 4801), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for i in range(16):                                   # the 16th launch after create is the tagged one
    dm.process_batch_device(batches[i % nb], audio)
dm.synchronize()
assert dm.kernel_name.startswith("fast-s"), dm.kernel_name
waves = ns * 8
raw = (C.c_uint64 * (32 * waves))()
lib = pkg.load_library(dev=True)
lib.sdrfm_dev_read_debug.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32]
assert lib.sdrfm_dev_read_debug(dm._h, raw, 32 * waves) == 0
t = np.frombuffer(raw, dtype=np.uint64).reshape(waves, 32).astype(np.int64)
cyc = t[:, :3 + 2 * (NB - 2) + 7]                 # entry, first lines, first body, (prio, body) x4, last, heads, audio (5 stamps)
names = ["first line wait", "first body"] + [x for b in range(1, NB - 1) for x in ("prio %d" % b, "body %d" % b)] + ["last body", "head pass", "audio: d -> LDS", "audio: taps + history", "audio: window reads issued", "audio: chains + results -> LDS", "audio: stores issued"]
d = np.diff(cyc, axis=1)
hw = t[:, 29]; xcc = hw & 0xf; hwid = hw >> 32
slot = hwid & 0xf; simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 0xf; se = (hwid >> 13) & 7; sh = (hwid >> 12) & 1
rt0, rt1 = t[:, 30], t[:, 31]
print("kernel:", dm.kernel_name, " waves:", waves)
print("%-32s %10s %10s %10s %10s" % ("phase", "mean cyc", "p10", "p90", "max"))
for i, n in enumerate(names):
    print("%-32s %10.0f %10.0f %10.0f %10.0f" % (n, d[:, i].mean(), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90), d[:, i].max()))
tot = cyc[:, -1] - cyc[:, 0]
print("%-32s %10.0f %10.0f %10.0f %10.0f" % ("whole wave", tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90), tot.max()))
dur_us = (rt1 - rt0) / 100.0
print("wave duration us: mean %.2f p10 %.2f p90 %.2f max %.2f  => shader clock %.3f GHz" % (dur_us.mean(), np.percentile(dur_us, 10), np.percentile(dur_us, 90), dur_us.max(), (tot / dur_us).mean() / 1e3))
for x in range(8):
    m = xcc == x
    if m.any():
        s0 = rt0[m].min()
        print("XCC%d: waves %4d  starts %.2f..%.2f us  ends %.2f..%.2f us (p50 %.2f)" % (x, m.sum(), 0.0, (rt0[m].max() - s0) / 100.0, (rt1[m].min() - s0) / 100.0, (rt1[m].max() - s0) / 100.0, (np.median(rt1[m]) - s0) / 100.0))
# residency: waves per (xcc, se, sh, cu, simd)
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
per_cu = np.bincount(key.astype(np.int64))
per_cu = per_cu[per_cu > 0]
print("CUs seen: %d; waves per CU: min %d max %d; waves starting within the first 1 us: %d of %d" % (len(per_cu), per_cu.min(), per_cu.max(), int(((rt0 - rt0.min()) < 100).sum()), waves))
late = (rt0 - rt0.min()) >= 100
if late.any():
    print("late waves: %d, start at %.1f..%.1f us" % (late.sum(), (rt0[late].min() - rt0.min()) / 100.0, (rt0[late].max() - rt0.min()) / 100.0))
ks = key * 4 + simd
per_simd = np.bincount(ks.astype(np.int64)); per_simd = per_simd[per_simd > 0]
print("waves per SIMD: histogram", np.bincount(per_simd).tolist())
print("hardware wave slots used: ", dict(zip(*[x.tolist() for x in np.unique(slot, return_counts=True)])))
for sl in np.unique(slot):
    m = slot == sl
    print("slot %d: first line wait mean %.0f, whole wave mean %.0f cycles, end p50 %.2f us" % (sl, d[m, 0].mean(), tot[m].mean(), (np.median(rt1[m]) - rt0.min()) / 100.0))
