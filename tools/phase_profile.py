#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of the fast kernel on the bench workload.
Runs the DEVELOPMENT library (csrc/libsdrfm_dev.so, `make -C stm32f7-rtlsdr_amd/csrc dev`): the product holds no instrumented kernel."""
import importlib, os, sys
os.environ["SDRFM_PHASE_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("stm32f7-rtlsdr_amd")
ns, nsamp = 256, 240000
h, g = pkg.default_config(int(os.environ.get("TAPS", "64")))
dm = pkg.FmDemod(pkg.FmConfig(fir_coeffs=h, audio_coeffs=g, n_streams=ns, dev_library=True))
iq = torch.from_numpy(np.tile(pkg.make_iq(16, nsamp), (16, 1))).cuda()
audio = torch.zeros((ns, 4801), dtype=torch.float32, device="cuda")
for _ in range(3):
    dm.process_batch_device(iq, audio)
dm.synchronize(); dm.phase_cycles()
reps = 20
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); t0.record()
for _ in range(reps):
    dm.process_batch_device(iq, audio)
dm.synchronize(); t1.record(); torch.cuda.synchronize()
import ctypes as C
raw = (C.c_uint64 * 560)()
dm._lib.sdrfm_debug_raw(dm._h, raw)
for x in range(8):
    sk = [int(raw[520 + 4 * x + i]) for i in range(4)]
    print('  XCC%d (us from its first wave start): last start %.2f, first end %.2f, last end %.2f' % (x, (sk[1]-sk[0])/100.0, (sk[2]-sk[0])/100.0, (sk[3]-sk[0])/100.0))
pc = dm.phase_cycles()
rt_ticks = pc["carry"] >> 32; pc["carry"] &= (1 << 32) - 1
wave_cyc = pc["waves"] >> 20; pc["waves"] &= (1 << 20) - 1
st = pc["subtiles"]
tot = sum(pc[k] for k in ("stage", "fir", "disc", "audio", "carry"))
print(dm.kernel_name, "| us/launch (instrumented, stream-sync'd): %.1f" % (t0.elapsed_time(t1) * 1e3 / reps))
print("waves/launch %d  subtiles/wave %.1f  cycles/subtile %.0f" % (pc["waves"] / reps, st / pc["waves"], tot / st))
for k in ("stage", "fir", "disc", "audio", "carry"):
    print("  %-6s %8.0f cycles/subtile  %5.1f%%" % (k, pc[k] / st, 100.0 * pc[k] / tot))
if rt_ticks:
    print("  whole wave: %.0f cycles, %.2f us real time  => shader clock %.2f GHz" % (wave_cyc / pc["waves"], rt_ticks / pc["waves"] / 100.0, wave_cyc / (rt_ticks / 100.0) / 1e3))
print("  prologue %8.0f cycles/wave; loop %8.0f cycles/wave" % (pc["prologue"] / pc["waves"], tot / pc["waves"]))
